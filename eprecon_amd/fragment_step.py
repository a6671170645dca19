"""One pass of the per-fragment hot path over one synthetic window, at BASELINE.json's configs.

Cfg2Step = config 2 ("9-frame 640x480 window, 96^3 coarse back_project + occupancy only"):
what NeuConNet.forward does before the sparse-conv U-Nets (models/neucon_network.py:239-369 of the
reference), with the back-projections on DENSE voxel grids — the upper-bound bandwidth case of
SURVEY.md section 8d:

  init    Occupancy_Initialization.forward on the dense 48^3 grid, min_view 2: 2D fusion convolutions
          (PyTorch-ROCm) -> view-variance volume -> BN / sparse ELAN / 3 residual SubM blocks /
          SubM(32->1) / BN                          (models/occupancy_initialization.py:61-182)
  select  sigmoid > 0.3 -> 2^3 OR-pool -> erode -> dilate x2 -> stage-0 coordinates
                                                    (models/neucon_network.py:264,298-318)
  bp24    Back_Project, dense 24^3 (interval 4), C=80 @ 30x40,   min_view 2   (stage 0)
  bp48    Back_Project, dense 48^3 (interval 2), C=40 @ 60x80,   min_view 0   (stage 1)
  bp96    Back_Project, dense 96^3 (interval 1), C=24 @ 120x160, min_view 0   (stage 2)

Inputs (both backbones' feature pyramids, projection matrices, voxel lists) are device-resident
before run() is called; weights are seeded random (no checkpoint exists in this environment).
"""
import os

import torch

from . import _lib
from . import back_project as BP
from . import grid_ops as GO
from . import synthetic as S
from .config import CH_IMG, CH_INIT_DOWN, N_VIEWS
from .occupancy_initialization import Occupancy_Initialization

LEVELS = [("bp24", 2, 4, 2), ("bp48", 1, 2, 0), ("bp96", 0, 1, 0)]  # name, proj level, interval, min_view


class Cfg2Step:
    def __init__(self, seed=0, device=None, height=480, width=640, n_vox=(96, 96, 96)):
        self.device = device or torch.device("cuda")
        self.seed = seed
        self.window = S.make_window(seed=seed, width=width, height=height, n_vox=n_vox)
        self.shapes = S.pyramid_shapes(height, width)
        dev = self.device
        t = lambda a: torch.from_numpy(a).to(dev)
        self.origin = t(self.window["vol_origin_partial"][None].copy())
        self.voxel_size = self.window["voxel_size"]
        self.krcam = [t(self.window["proj_matrices"][:, l][:, None].copy()) for l in range(3)]
        # backbone #2 pyramid (surface reconstruction) as [V,B,C,H,W] stacks per level
        self.feats = [t(S.make_features(1000 * seed + 10 + l, N_VIEWS, self.shapes[l])) for l in range(3)]
        # backbone #1 pyramid (initialisation) in the reference's list-of-views form
        f1 = [t(S.make_features(1000 * seed + 20 + l, N_VIEWS, self.shapes[l])) for l in range(3)]
        self.features_init = [[f1[l][v] for l in range(3)] for v in range(N_VIEWS)]
        # dense x-major rasters of the fragment volume (what generate_grid yields), tagged as such
        self.coords = {iv: BP.mark_dense(t(S.dense_coords(n_vox, iv)), [n // iv for n in n_vox], iv) for iv in (4, 2, 1)}
        self.shape_init = tuple(n // 2 for n in n_vox)
        torch.manual_seed(1234)
        self.init_net = Occupancy_Initialization(CH_IMG, CH_INIT_DOWN, N_VIEWS).to(dev)
        self.init_net.train()  # the reference tests in train mode (main.py:357)
        self.last = {}
        # defer_reads: run() queues step k and returns step k - 1's outputs — the host reads of a step's counts (stage-0
        # selection, valid voxels of the three levels) happen after the NEXT step is queued, so the GPU does not idle at the
        # step boundary while the host starts issuing; flush() finishes the last step.  Off: run() returns its own outputs.
        self.defer_reads = False
        self._inflight = None
        # the Back_Project levels are queued behind the variance volume, in front of the host's wait for its count
        # (levels_inside = False: at the start of the step, round 2's order: 1.67-1.70 ms against 1.60 ms)
        self.levels_inside = True
        self.profile_dominant = False  # bench.py: time the dense 96^3 gather with the library's event pair
        # (the three levels on their own stream: 1.57 ms, but the timed gather stretches to 142 us next to the 2D stack)
        self.bp_side_stream = False
        self._bp_stream = None

    @torch.no_grad()
    def run(self):
        """One pass of the cfg2 hot path: initialisation branch (2D stack, variance volume, submanifold stack, stage-0
        selection) and the three dense Back_Project levels, which do not depend on it and are queued where the branch has to
        wait for its valid-voxel count (_issue).  Same stream: the kernels still run one after the other, only the host gaps
        disappear."""
        issued = self._issue()
        if not self.defer_reads:
            return self._finish(issued)
        prev, self._inflight = self._inflight, issued
        return self._finish(prev) if prev is not None else {}

    def flush(self):
        """deferred mode: read the counts of the step still in flight"""
        if self._inflight is not None:
            prev, self._inflight = self._inflight, None
            return self._finish(prev)
        return self.last

    def _finish(self, issued):
        out, pending, select = issued
        if select is not None:
            out["stage0_coords"], _ = select.result()
        for name in pending:
            out[name] = pending[name].result()
        self.last = out
        return out

    def _issue(self):
        out = {}
        pending = {}
        main = torch.cuda.current_stream(self.device)
        side = self._bp_stream if self.bp_side_stream else main

        def queue_levels():
            # The three dense Back_Project levels do not depend on the initialisation branch.  They are queued without a host
            # round trip between them (run_async) at the point where the initialisation branch has to wait for its valid-voxel
            # count: the GPU works through them while the host reads the count and starts issuing the submanifold stack.
            if side is not main:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                # bp_side_stream: the three levels on their own stream, concurrent with the initialisation branch
                for name, lvl, interval, mv in LEVELS:
                    if self.profile_dominant and name == "bp96":
                        _lib.load().eprecon_profile_enable(2)  # one-shot: bracket this level's gather kernel only
                    pending[name] = BP.run_async(self.coords[interval], self.origin, self.voxel_size, self.feats[lvl],
                                                 self.krcam[lvl], mv)

        if not self.levels_inside:
            queue_levels()
        init = self.init_net(self.coords[2], self.origin, self.voxel_size, self.features_init,
                             self.krcam[1], self.shape_init, 1, 2, between=queue_levels if self.levels_inside else None)
        out["init"] = init
        select = None
        if init is not None:
            select = GO.init_select_async(init[0], init[1], 1, dim=self.shape_init[0] // 2, cell=4)
        if side is not main:
            main.wait_stream(side)
        return out, pending, select

    def dominant_kernel_ms(self, lib):
        """elapsed ms of the last bp_gather launch (the dense 96^3 level), from the library's
        HIP event pair recorded on the launch stream"""
        return float(lib.eprecon_profile_gather_ms())

    def dominant_kernel_bytes(self):
        """algorithmic bytes of the bp_gather launch on the 96^3 level (kernels):
        coords in 16 N + maps once 4 V C H W + rows out n_valid (4 C + 16)"""
        c, h, w = self.shapes[0]
        n = self.coords[1].shape[0]
        nv = self.last["bp96"]["n_valid"] if self.last.get("bp96") else n
        return 16 * n + 4 * N_VIEWS * c * h * w + nv * (4 * c + 16)

    def dominant_kernel_l1_bandwidth(self, launch_ms):
        """What the gather asks of the vector L1s (the limiter of this kernel): every valid voxel reads, per view
        it is visible in, four bilinear taps of C floats — an UPPER bound on the requested bytes is n_valid x V x 4 x 4 C (all V
        views visible) — against 64 B per clock per CU x 256 CUs x 2.4 GHz."""
        if not launch_ms:
            return None
        c = self.shapes[0][0]
        n = self.coords[1].shape[0]
        nv = self.last["bp96"]["n_valid"] if self.last.get("bp96") else n
        req = float(nv) * N_VIEWS * 4 * 4 * c
        peak = 64.0 * 256 * 2.4e9 / 1e12
        ach = req / (launch_ms * 1e-3) / 1e12
        return {"requested_bytes": req, "requested_note": "upper bound: every valid voxel visible in all views (4 taps x 4 C bytes per view)",
                "achieved_TBps": ach, "peak_TBps": peak, "frac": ach / peak}

    def describe(self):
        return {"workload": "cfg2: one 9-view 640x480 window, 96^3 FBV: Occupancy_Initialization on the dense "
                            "48^3 grid (2D fusion convs + variance volume + submanifold stack) + stage-0 "
                            "selection + Back_Project on dense 24^3/48^3/96^3 (C=80/40/24)",
                "views": N_VIEWS, "image": "640x480", "n_vox": list(self.window["n_vox"]),
                "stages": ["occupancy_init48", "init_select"] + [l[0] for l in LEVELS],
                "weights": "seeded random", "fragments_per_step_per_gpu": 1,
                "host_reads": ("counts of step k read after step k + 1 is queued (flushed inside the timed region)"
                               if self.defer_reads else "every step reads its own counts before returning")}


WORKLOAD_SEED = 20240  # np.random state of every calibration and workload forward (the reference's sub-sampling draws from it)


def seed_subsampling(k=0):
    """The reference sub-samples over-full levels with np.random.choice (models/neucon_network.py:478-484).  Synthetic
    workloads fix that generator's state before every forward, so calibration and timed runs draw the SAME subsets and the
    sparsity of a workload is a controlled regime, not an accident of how many forwards ran before it."""
    import numpy as np
    np.random.seed(WORKLOAD_SEED + int(k))


@torch.no_grad()
def calibrate_occupancy_heads(net, features, features_occ_pano=None, inputs=None, keep_fraction=(0.45, 0.35, 0.25)):
    """Random-init occupancy heads may classify (almost) every voxel the same way, which trips the
    reference's `< 500 occupied voxels` early return.  There is no trained checkpoint in this
    environment, so benchmarks and end-to-end tests rescale each stage's occupancy logit
    (occ_preds[i].linear3: logit' = (logit - q) / sigma, q the (1 - keep_fraction[i]) quantile and sigma the standard
    deviation over the stage's voxels) such that `keep_fraction[i]` of the stage's voxels pass `occ > 0` — the sparsity
    regime "random ~50 %" of SURVEY.md section 8d.  Unit-variance logits make the fraction insensitive to the last
    bits of the weights (an Adam step, another GRU state): seeded-random heads give a logit spread of ~1e-2, where a
    bias set on one forward kept 66-71 % instead of 25 % on the next.
    `features` may also be a list of (features, features_occ_pano, inputs) fragments of ONE scene: they are run in
    sequence (live GRU state) and the statistics pooled.  One pass over the fragments per stage; the GRU state is
    reset afterwards."""
    frags = features if features_occ_pano is None else [(features, features_occ_pano, inputs)]
    old_trace = net.trace
    # the forwards below run with uncalibrated heads and trip the reference's guards by design ("no valid points", "exceed
    # too many points"): their warnings are tagged so that a log line from a TIMED forward is attributable
    from . import neucon_network as _NN
    old_tag, _NN.WARN_TAG = _NN.WARN_TAG, "[calibration] "
    try:
        _calibrate(net, frags, keep_fraction)
    finally:
        _NN.WARN_TAG = old_tag
        net.gru_fusion.scene_name = [None, None, None]
        net.trace = old_trace


def _calibrate(net, frags, keep_fraction):
    for i in range(net.cfg.N_LAYER):
        net.gru_fusion.scene_name = [None, None, None]
        occs = []
        for k, (f1, f2, inp) in enumerate(frags):
            net.trace = []
            seed_subsampling(k)
            net(f1, f2, inp, {})
            rec = {t["stage"]: t for t in net.trace}.get(f"heads{i}")
            if rec is not None:
                occs.append(rec["occ"][:, 0].float())
        if not occs:
            raise RuntimeError(f"stage {i} was not reached while calibrating")
        occ = torch.cat(occs)
        if occ.numel() > 4_000_000:  # torch.quantile's input limit is 16 M elements; a strided sample is plenty
            occ = occ[:: occ.numel() // 4_000_000 + 1]
        q = torch.quantile(occ, 1.0 - keep_fraction[i])
        sigma = occ.std().clamp_min(1e-12)
        lin = net.occ_preds[i].linear3
        lin.bias.sub_(q).div_(sigma)
        lin.weight.div_(sigma)


class Cfg4Step:
    """Configs 3/4: the whole NeuConNet.forward (occupancy initialisation, three coarse-to-fine levels
    of Back_Project + SPVCNN + GRU fusion + heads, panoptic inputs) over `n_fragments` consecutive
    windows of one scene (camera arc advanced 0.32 m per fragment, persistent GRU map).
    One step = one fragment; the scene restarts every `n_fragments` steps."""

    def __init__(self, seed=0, device=None, height=480, width=640, n_fragments=4, rank=0, world=1, force_exchange=False,
                 pipeline=False):
        from .config import ModelCfg
        from .neucon_network import NeuConNet
        self.device = device or torch.device("cuda")
        self.seed, self.n_fragments = seed, n_fragments
        torch.manual_seed(4321)
        self.net = NeuConNet(ModelCfg()).to(self.device)
        self.net.train()
        self.frags = []
        for k in range(n_fragments):
            # make_window keeps vol_origin fixed and snaps vol_origin_partial to the advanced arc,
            # so the fragments of one scene share the global origin and overlap by 2/3
            # with `world` ranks the fragments of the scene are dealt round-robin: rank r owns
            # fragments r, r + world, ... and exchanges boundary voxels with the others every step
            kk = k * world + rank
            # the camera arc advances 0.32 m per fragment on one GPU (2/3 overlap); with more ranks the same 0.96 m of
            # the synthetic room is covered in proportionally smaller steps, so no fragment leaves the geometry
            advance = min(0.32, 0.96 / max(n_fragments * world - 1, 1))
            w = S.make_window(seed=seed * 100 + kk, width=width, height=height, advance=advance * kk)
            f1, f2, inp = S.make_model_inputs([w], feat_seed=seed * 100 + kk, scene=f"scene{seed:04d}")
            self.frags.append((S.to_device(f1, self.device), S.to_device(f2, self.device),
                               S.to_device(inp, self.device)))
        # the occupancy biases are calibrated on rank 0's first fragment and broadcast: every rank must run the
        # SAME weights (a per-rank calibration would give the ranks of one scene different networks)
        if world > 1:
            import torch.distributed as dist
            if rank == 0:
                calibrate_occupancy_heads(self.net, self.frags)
            for p_ in self.net.parameters():
                dist.broadcast(p_.data, 0)
            from .sparse import clear_packed_weights
            clear_packed_weights(self.net)     # (writes through .data do not bump the version the weight caches key on)
        else:
            calibrate_occupancy_heads(self.net, self.frags)
        # EPRECON_FORCE_EXCHANGE=1: run the boundary all-gather even at world size 1 (exercises the RCCL path on one GPU)
        self.net.distributed_exchange = world > 1 or os.environ.get("EPRECON_FORCE_EXCHANGE", "0") == "1" or force_exchange
        self._pending = None
        self.pipeline = False
        self.set_pipeline(pipeline)
        self.k = 0
        self.last = None
        self.voxels = []  # finest-level voxel count of every fragment run so far
        self.early_returns = 0
        # multi-GPU: a rank that raised would leave the others waiting in the next collective; count instead
        self.raise_on_early_return = world == 1

    def set_pipeline(self, pipeline):
        """False: every fragment is complete when run() returns (the reference's contract).  True: the panoptic branch of
        fragment k is issued by a worker thread on its own stream and overlaps the front of fragment k + 1, its host-side
        post-processing is finished one step later (NeuConNet.panoptic_stream / panoptic_worker: a THROUGHPUT mode that
        changes the outputs contract — `panoptic_finish`).  "inline": the side stream without the worker thread."""
        self.flush()
        if self.net.panoptic_worker is not None:
            self.net.panoptic_worker.shutdown(wait=True)
        self.net.panoptic_stream = self.net.panoptic_worker = None
        self.pipeline = pipeline
        if pipeline:
            self.net.panoptic_stream = _lib.side_stream(self.device, _lib.SIDE_PANOPTIC)
            if pipeline != "inline":
                from concurrent.futures import ThreadPoolExecutor
                self.net.panoptic_worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix="eprecon-panoptic")

    @torch.no_grad()
    def run(self):
        if self.k == 0:
            self.net.gru_fusion.scene_name = [None, None, None]
        f1, f2, inp = self.frags[self.k]
        seed_subsampling(self.k)
        self.last, _ = self.net(f1, f2, inp, {})
        self.flush()                                           # the PREVIOUS fragment's panoptic branch (long finished)
        self._pending = self.last.get("panoptic_finish")
        if "coords" not in self.last or ("panoptic_levels" not in self.last and "panoptic_finish" not in self.last):
            # a data-dependent early return of NeuConNet.forward (< 500 occupied voxels, no valid points, over the
            # cap) would otherwise be timed as a very fast fragment
            if self.raise_on_early_return:
                raise RuntimeError(f"fragment {self.k}: NeuConNet.forward returned before the finest level "
                                   f"(outputs: {sorted(self.last)})")
            self.early_returns += 1
        else:
            self.voxels.append(int(self.last["coords"].shape[0]))
        self.k = (self.k + 1) % self.n_fragments
        return self.last

    def flush(self):
        """finish the deferred panoptic post-processing of the last fragment (pipelined mode; no-op otherwise)"""
        if self._pending is not None:
            self._pending()
            self._pending = None

    @torch.no_grad()
    def run_cfg3(self):
        """BASELINE.json configs[2]: ONE fragment through the full 3-level coarse-to-fine TSDF path with an EMPTY
        global map (scene restarted) and without the panoptic decoder"""
        self.flush()
        dec, self.net.panoptic = self.net.panoptic, None
        try:
            self.net.gru_fusion.scene_name = [None, None, None]
            f1, f2, inp = self.frags[0]
            seed_subsampling(0)
            out, _ = self.net(f1, f2, inp, {})
        finally:
            self.net.panoptic = dec
            self.net.gru_fusion.scene_name = [None, None, None]
            self.k = 0
        if "coords" not in out:
            raise RuntimeError("cfg3: NeuConNet.forward returned before the finest level")
        return out

    def describe(self):
        return {"workload": f"cfg4: NeuConNet.forward over {self.n_fragments} sequential 9-view 640x480 fragments "
                            "(occupancy init, 3 x [Back_Project, SPVCNN, GRU fusion, heads], panoptic inputs), "
                            "96^3 FBV, persistent sparse global map",
                "views": N_VIEWS, "image": "640x480", "weights": "seeded random, occupancy heads calibrated to "
                "45/35/25 % keep", "fragments_per_step_per_gpu": 1,
                "pipelined": "panoptic branch of fragment k on its own stream, overlapping fragment k + 1" if self.pipeline else "no"}


class E2EStep:
    """The drop-in boundary itself: NeuralRecon.forward(inputs, save_mesh=False, training=False) the way main.py:430-436
    calls it at test time (models/neuralrecon.py:46-86) — image normalisation, BOTH MnasMulti backbones on the nine 640x480
    images, the whole HIP 3D path (NeuConNet.forward incl. the panoptic decoder) and fuse_to_global — over `n_fragments`
    consecutive windows of one scene.  One step = one fragment; nothing is pipelined across fragments (outputs carry
    `panoptic_info` and the fused scene state when the call returns, the reference's contract)."""

    def __init__(self, seed=0, device=None, height=480, width=640, n_fragments=4):
        import numpy as np
        from .config import ModelCfg
        from .neuralrecon import NeuralRecon
        self.device = device or torch.device("cuda")
        self.n_fragments = n_fragments
        torch.manual_seed(4321)
        self.model = NeuralRecon(ModelCfg()).to(self.device)
        self.model.train()     # main.py:357
        self.inputs = []
        for k in range(n_fragments):
            w = S.make_window(seed=seed * 100 + k, width=width, height=height, advance=0.32 * k)
            _, _, inp = S.make_model_inputs([w], feat_seed=seed * 100 + k, scene=f"scene{seed:04d}")
            rng = np.random.default_rng(seed * 100 + k + 7)
            inp["imgs"] = rng.random((1, N_VIEWS, 3, height, width), dtype=np.float32) * 255.0   # 0..255 like main.py:113
            self.inputs.append(S.to_device(inp, self.device))
        with torch.no_grad():
            frags = []
            for inp in self.inputs:
                norm = [self.model.normalizer(img) for img in torch.unbind(inp["imgs"], 1)]
                frags.append((self.model.backbone2d.forward_views(norm), self.model.backbone_occ_pano.forward_views(norm), inp))
            calibrate_occupancy_heads(self.model.neucon_net, frags)
        self.k = 0
        self.last = None
        self.voxels = []
        self.early_returns = 0

    @torch.no_grad()
    def run(self):
        if self.k == 0:   # scene restart: both persistent maps (GRU features, fused TSDF / instances)
            self.model.neucon_net.gru_fusion.scene_name = [None, None, None]
            self.model.fuse_to_global.scene_name = [None, None, None]
        seed_subsampling(self.k)
        self.last, _ = self.model(self.inputs[self.k], save_mesh=False, training=False)
        if "coords" not in self.last or "panoptic_info" not in self.last:
            self.early_returns += 1
        else:
            self.voxels.append(int(self.last["coords"].shape[0]))
        self.k = (self.k + 1) % self.n_fragments
        return self.last

    def describe(self):
        return {"workload": f"NeuralRecon.forward(inputs, save_mesh=False, training=False) over {self.n_fragments} sequential "
                            "9-view 640x480 fragments of one scene: normalisation + 2 x MnasMulti on the images + "
                            "NeuConNet.forward (panoptic decoder included) + fuse_to_global; images resident in HBM"}


class TrainStep:
    """One optimisation step of the 3D path the way main.py:297-313 takes it: NeuConNet.forward with autograd on one
    fragment (recording operators of eprecon_amd/autograd.py), the LW-weighted total loss (models/neuralrecon.py:76-84),
    loss.backward(), clip_grad_norm_(1.0), Adam (main.py:164: lr, betas (0.9, 0.999)).  With `world` > 1 the network is
    wrapped in DistributedDataParallel over RCCL exactly as main.py:155-162 does (broadcast_buffers=False,
    find_unused_parameters=True): one gradient all-reduce per step, bucketed by torch.

    The image pyramids are leaf tensors standing in for the two 2D backbones (their gradient is what the backbones
    would receive); the scene map is reset before every step so that each step sees the same fragment."""

    LW = (1.0, 0.8, 0.64, 1.2)       # config/train.yaml:44

    def __init__(self, seed=0, device=None, height=480, width=640, lr=1e-4, rank=0, world=1, batch=1):
        """batch: fragment windows per step and rank (consecutive windows of the scene; the reference trains at BATCH_SIZE 4,
        config/train.yaml:2 — every stage of the forward loops over the batch elements, the BatchNorms see them all)"""
        from .config import ModelCfg
        from .neucon_network import NeuConNet
        self.device = device or torch.device("cuda")
        self.batch = batch
        torch.manual_seed(4321)
        self.net = NeuConNet(ModelCfg()).to(self.device)
        self.net.train()
        ws = [S.make_window(seed=seed * 100 + rank * batch + b, width=width, height=height, advance=0.32 * (rank * batch + b))
              for b in range(batch)]
        f1, f2, inp = S.make_model_inputs(ws, feat_seed=seed * 100 + rank, scene=f"scene{seed:04d}", panoptic=True)
        self.f1, self.f2, self.inputs = (S.to_device(x, self.device) for x in (f1, f2, inp))
        calibrate_occupancy_heads(self.net, self.f1, self.f2, self.inputs)
        for views in (self.f1, self.f2):
            for levels in views:
                for t in levels:
                    t.requires_grad_()
        self.model = self.net
        if world > 1:
            import torch.distributed as dist
            from torch.nn.parallel import DistributedDataParallel
            for p_ in self.net.parameters():
                dist.broadcast(p_.data, 0)
            from .sparse import clear_packed_weights
            clear_packed_weights(self.net)
            self.model = DistributedDataParallel(self.net, device_ids=[self.device.index], output_device=self.device.index,
                                                 broadcast_buffers=False, find_unused_parameters=True)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr, betas=(0.9, 0.999))
        self.last = None
        self.steps = 0
        self.early_returns = 0      # steps whose forward returned before the finest level / the panoptic criterion
        self.voxels = []            # finest-level voxel count of every full step
        # a truncated step (the reference's guards, models/neucon_network.py:463-497) is not an optimisation step of this
        # workload: raise on one GPU; under DDP a raising rank would strand the others in the gradient all-reduce -> count
        self.raise_on_early_return = world == 1

    def describe(self):
        return {"workload": f"one optimisation step on {'a batch of ' + str(self.batch) + ' consecutive' if self.batch > 1 else 'one'} 9-view "
                            f"640x480 fragment{'s' if self.batch > 1 else ''} (empty scene map): NeuConNet.forward under "
                            "autograd, TSDF / occupancy losses of the three levels + panoptic set criterion, backward through "
                            "the HIP operators, clip_grad_norm_(1.0), Adam; image pyramids are leaf tensors",
                "parallelism": "DistributedDataParallel over RCCL, one fragment per rank" if self.model is not self.net else "single"}

    def loss(self):
        self.net.gru_fusion.scene_name = [None, None, None]
        seed_subsampling(0)
        outputs, loss_dict = self.model(self.f1, self.f2, self.inputs, {})
        total = 0
        for i, (k, v) in enumerate(loss_dict.items()):
            total = total + v * self.LW[min(i, len(self.LW) - 1)]
        loss_dict["total_loss"] = total
        return outputs, loss_dict

    def run(self):
        self.optimizer.zero_grad(set_to_none=True)
        outputs, loss_dict = self.loss()
        self.steps += 1
        if "coords" not in outputs or "panoptic_loss" not in loss_dict:
            # the forward hit one of the reference's data-dependent early returns: the levels behind it and the set
            # criterion did not run, so this is NOT the step the workload describes
            if self.raise_on_early_return:
                raise RuntimeError(f"train step {self.steps}: NeuConNet.forward returned early (losses: {sorted(loss_dict)})")
            self.early_returns += 1
        else:
            self.voxels.append(int(outputs["coords"].shape[0]))
        loss_dict["total_loss"].backward()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
        self.optimizer.step()
        # every weight has changed: all operand-order copies rebuilt by ONE launch here instead of one per layer at its next use
        if os.environ.get("EPRECON_TRAIN_REPACK", "1") == "1":
            from .sparse import repack_registered
            repack_registered()
        self.last = {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in loss_dict.items()}
        return self.last
