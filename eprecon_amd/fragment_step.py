"""One pass of the per-fragment hot path over one synthetic window, at BASELINE.json's configs.

Cfg2Step = config 2 ("9-frame 640x480 window, 96^3 coarse back_project + occupancy only"):
the stages NeuConNet.forward runs before the sparse-conv U-Nets (models/neucon_network.py:239-369
of the reference), on dense voxel grids — the upper-bound bandwidth case of SURVEY.md section 8d:

  init   view-variance volume of the fused 32-channel 60x80 maps on the dense 48^3 grid, min_view 2
         (models/occupancy_initialization.py:79-128)
  bp24   Back_Project, dense 24^3 (interval 4), C=80 @ 30x40,   min_view 2   (stage 0)
  bp48   Back_Project, dense 48^3 (interval 2), C=40 @ 60x80,   min_view 0   (stage 1)
  bp96   Back_Project, dense 96^3 (interval 1), C=24 @ 120x160, min_view 0   (stage 2)

Inputs are device-resident before run() is called; every stage goes through the C ABI.
"""
import torch

from . import back_project as BP
from . import synthetic as S
from .config import CH_INIT_DOWN, N_VIEWS

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
        self.feats = [t(S.make_features(1000 * seed + 10 + l, N_VIEWS, self.shapes[l])) for l in range(3)]
        c1, h1, w1 = self.shapes[1]
        self.feats_init = t(S.make_features(1000 * seed + 20, N_VIEWS, (CH_INIT_DOWN, h1, w1)))
        self.coords = {iv: t(S.dense_coords(n_vox, iv)) for iv in (4, 2, 1)}
        self.last = {}

    def run(self):
        out = {}
        out["init"] = BP.view_variance(self.coords[2], self.origin, self.voxel_size, self.feats_init,
                                       self.krcam[1], 2)
        for name, lvl, interval, mv in LEVELS:
            out[name] = BP.run(self.coords[interval], self.origin, self.voxel_size, self.feats[lvl],
                               self.krcam[lvl], mv)
        self.last = out
        return out

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

    def describe(self):
        return {"workload": "cfg2: one 9-view 640x480 window, 96^3 FBV: view-variance volume on dense 48^3 "
                            "(C=32 @60x80) + Back_Project on dense 24^3/48^3/96^3 (C=80/40/24)",
                "views": N_VIEWS, "image": "640x480", "n_vox": list(self.window["n_vox"]),
                "stages": ["init_variance48"] + [l[0] for l in LEVELS], "fragments_per_step_per_gpu": 1}
