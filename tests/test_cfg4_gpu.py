"""GPU: BASELINE.json configs[3] at full size — 4 sequential 9-view 640x480 fragments of one scene through
NeuConNet.forward with the persistent GRU map and the panoptic head, checked stage by stage against the CPU
oracle on the inputs the HIP path fed each stage (indices bit-exact, features within 1e-3):
  * every fragment reaches the finest level;
  * GRU fusion: union order / coordinates, ground-truth targets and the map after update_map bit-exact
    against oracle/gru_fusion.py driven through all four fragments; fused features against oracle ConvGRUs
    (reference-literal convr); fragment k's union contains the part of fragment k-1's map inside its FBV;
  * SPVCNN + heads on fragments 0 and 2 (the oracle takes ~20 s per fragment at these sizes);
  * the `np.random.choice` drop of models/neucon_network.py:477-484 with the same numpy seed on both sides."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import gru_fusion as OGF  # noqa: E402
from oracle import neucon as ONC  # noqa: E402
from oracle import pointvoxel as PV  # noqa: E402
from oracle import spvcnn as ON  # noqa: E402

TOL = 1e-3
N_FRAG = 4


def npy(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def run4():
    from eprecon_amd.fragment_step import Cfg4Step
    np.random.seed(3)
    step = Cfg4Step(seed=0, device=torch.device("cuda"), n_fragments=N_FRAG)
    net = step.net
    frames = []
    for k in range(N_FRAG):
        net.trace = []
        out = step.run()                     # raises if the forward returns before the finest level
        maps = [(npy(net.gru_fusion.global_volume[s].C), npy(net.gru_fusion.global_volume[s].F)) for s in range(3)]
        tmaps = [(npy(net.gru_fusion.target_tsdf_volume[s].C), npy(net.gru_fusion.target_tsdf_volume[s].F))
                 for s in range(3)]
        frames.append({"trace": {t["stage"]: {a: (npy(b) if torch.is_tensor(b) else b) for a, b in t.items()}
                                 for t in net.trace if t["stage"] != "panoptic"},
                       "out": out, "maps": maps, "tmaps": tmaps})
    net.trace = None
    sd = {k: npy(v) for k, v in net.state_dict().items()}
    inputs = [{k: (npy(v) if torch.is_tensor(v) else v) for k, v in fr[2].items() if k not in ("tsdf_list", "occ_list")}
              for fr in step.frags]
    gts = [([npy(t) for t in fr[2]["tsdf_list"]], [npy(t) for t in fr[2]["occ_list"]]) for fr in step.frags]
    return {"step": step, "frames": frames, "sd": sd, "inputs": inputs, "gts": gts}


def test_every_fragment_reaches_the_finest_level(run4):
    for k, fr in enumerate(run4["frames"]):
        out = fr["out"]
        n = out["coords"].shape[0]
        assert n > 500 and out["tsdf"].shape == (n, 1), k
        assert out["panoptic_out"][0]["pred_masks"].shape == (1, 80, n)
        seg = out["panoptic_info"][0]["panoptic_seg"][0]
        assert seg.shape[0] == n and seg.dtype == torch.int32
        assert all(f"heads{i}" in fr["trace"] and f"gru{i}" in fr["trace"] for i in range(3))
    # the scene grows: later fragments fuse with a non-empty map
    sizes = [fr["maps"][2][0].shape[0] for fr in run4["frames"]]
    assert all(b > a for a, b in zip(sizes, sizes[1:])), sizes


@pytest.mark.parametrize("scale", [0, 1, 2])
def test_gru_fusion_over_four_fragments(run4, scale):
    sd, frames = run4["sd"], run4["frames"]
    interval = 2 ** (2 - scale)
    dim, vres = 96 // interval, 0.04 * interval
    ch_all, cv = (176, 88, 48)[scale], (96, 48, 24)[scale]
    state = OGF.ScaleState(ch_all, run4["inputs"][0]["vol_origin"][0])
    worst = 0.0
    for k, fr in enumerate(frames):
        t = fr["trace"][f"gru{scale}"]
        inp = run4["inputs"][k]
        tsdf_l, occ_l = run4["gts"][k]
        lvl = 2 - scale
        prev_c = state.C.copy()

        def fuse(gvals, vals, updated, rel, inp=inp):
            c4 = np.concatenate([np.zeros((len(updated), 1), np.int32), (updated * interval).astype(np.int32)], 1)
            pts = PV.aligned_coords(c4, inp["vol_origin_partial"], 0.04, inp["world_to_aligned_camera"])
            fv = ON.convgru(sd, f"gru_fusion.fusion_nets_voxel.{scale}", gvals[:, :cv], vals[:, :cv], pts, 1, vres)
            fi = ON.convgru(sd, f"gru_fusion.fusion_nets_img.{scale}", gvals[:, cv:], vals[:, cv:], pts, 1, vres)
            return np.concatenate([fv, fi], 1)

        r = OGF.fuse_fragment(state, t["coords_in"], t["feat_in"], inp["vol_origin_partial"][0], tsdf_l[lvl][0],
                              occ_l[lvl][0], interval, dim, fuse=fuse)
        assert np.array_equal(t["coords"][:, 1:], r["updated"] * interval), (k, scale)   # raster-ordered union
        assert not t["coords"][:, 0].any()
        assert np.array_equal(t["tsdf_target"], r["tsdf_target"])
        err = np.abs(t["feat_all"] - r["fused"]).max()
        worst = max(worst, err)
        assert err < TOL, (k, scale, err)
        # fragment k's union contains every voxel of fragment k-1's map that lies inside its bounding volume
        local = prev_c - r["rel"]
        inside = local[((local >= 0) & (local < dim)).all(1)]
        got = {tuple(v) for v in r["updated"]}
        assert all(tuple(v) in got for v in inside[:: max(1, len(inside) // 2000)])
        if k > 0:
            assert len(inside) > 0
        # the HIP map after update_map == the oracle's (same order), then follow the HIP values so that
        # rounding differences do not accumulate across fragments
        assert np.array_equal(fr["maps"][scale][0], state.C)
        assert np.abs(fr["maps"][scale][1] - state.F).max() < TOL
        assert np.array_equal(fr["tmaps"][scale][0], state.tC) and np.array_equal(fr["tmaps"][scale][1], state.tF)
        state.F = fr["maps"][scale][1].copy()
    print(f"scale {scale}: max |fused - oracle| over {len(frames)} fragments = {worst:.2e}")


@pytest.mark.parametrize("k", [0, 2])
@pytest.mark.parametrize("i", [0, 1, 2])
def test_spvcnn_and_heads(run4, k, i):
    tr, inp, sd = run4["frames"][k]["trace"], run4["inputs"][k], run4["sd"]
    t = tr[f"spvcnn{i}"]
    r, feat = ONC.spvcnn_stage(sd, i, t["coords"], t["feat_in"], inp["vol_origin_partial"],
                               inp["world_to_aligned_camera"])
    assert np.array_equal(t["r_coords"], r)
    err = np.abs(t["feat_out"] - feat).max()
    assert err < TOL, (err, np.abs(feat).max())
    h = tr[f"heads{i}"]
    tsdf, occ, occupancy = ONC.heads_stage(sd, i, h["feat"])
    assert np.abs(h["tsdf"] - tsdf).max() < TOL and np.abs(h["occ"] - occ).max() < TOL
    flips = h["occupancy"] != occupancy
    assert np.all(np.abs(occ[flips, 0]) < 1e-4)


def test_random_drop_branch_is_seeded_like_the_reference(run4):
    """num_batch in (cap, 1.5 cap]: `np.random.choice(num_batch, num_batch - cap, replace=False)` rows of the
    occupied voxels are dropped (models/neucon_network.py:477-484) — same numpy seed, same rows"""
    step = run4["step"]
    net = step.net
    f1, f2, inp = step.frags[0]
    occ0 = run4["frames"][0]["trace"]["heads0"]["occ"][:, 0]
    num = int((occ0 > 0).sum())
    old = list(net.cfg.TRAIN_NUM_SAMPLE)
    cap = int(num / 1.25)
    try:
        net.cfg.TRAIN_NUM_SAMPLE[0] = cap
        net.gru_fusion.scene_name = [None, None, None]
        net.trace = []
        np.random.seed(1234)
        with torch.no_grad():
            net(f1, f2, inp, {})
        rec = {t["stage"]: t for t in net.trace}["heads0"]
    finally:
        net.cfg.TRAIN_NUM_SAMPLE[:] = old
        net.trace = None
        net.gru_fusion.scene_name = [None, None, None]
    occ = npy(rec["occ"])[:, 0]
    before = occ > 0
    assert int(before.sum()) == num
    np.random.seed(1234)
    choice = np.random.choice(num, num - cap, replace=False)
    expect = before.copy()
    expect[np.nonzero(before)[0][choice]] = False
    assert int(expect.sum()) == cap
    assert np.array_equal(npy(rec["occupancy"]), expect)


def test_exchange_path_is_a_no_op_at_world_size_1():
    """the multi-GPU schedule's code path (RCCL process group, GRUFusion.exchange_boundaries: selection kernels on the map
    handles -> collectives -> stamp bookkeeping) run in a single-rank group must leave every result bit-identical"""
    import socket
    import torch.distributed as dist
    from eprecon_amd.fragment_step import Cfg4Step
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda")
    np.random.seed(3)
    plain = Cfg4Step(seed=0, device=dev, n_fragments=2)
    ref = [plain.run() for _ in range(2)]
    ref = [(npy(o["coords"]), npy(o["tsdf"])) for o in ref]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        np.random.seed(3)
        step = Cfg4Step(seed=0, device=dev, n_fragments=2)
        step.net.distributed_exchange = True
        for k in range(2):
            out = step.run()
            assert np.array_equal(npy(out["coords"]), ref[k][0]) and np.array_equal(npy(out["tsdf"]), ref[k][1])
        xch = step.net.gru_fusion._xchg
        assert xch is not None and xch.collectives == 4     # boxes + counts per fragment; nothing to send -> no payload all-gather
        gmap = step.net.gru_fusion.global_volume[2]          # the stamps are a column of the map handle on the GPU
        assert gmap.size > 0 and int((gmap.stamps() > 0).sum()) == gmap.size
    finally:
        dist.destroy_process_group()


def test_pipelined_panoptic_branch_gives_identical_fragments():
    """NeuConNet.panoptic_stream (the panoptic branch of fragment k on its own stream, overlapping fragment k + 1, its
    post-processing finished one step later) changes WHEN things run, not what they compute"""
    from eprecon_amd.fragment_step import Cfg4Step
    ref = Cfg4Step(seed=3, n_fragments=3)
    expected = []
    for _ in range(3):
        out = ref.run()
        expected.append({"coords": out["coords"].clone(), "tsdf": out["tsdf"].clone(),
                         "logits": out["panoptic_out"][0]["pred_logits"].clone(), "masks": out["panoptic_out"][0]["pred_masks"].clone(),
                         "seg": out["panoptic_info"][0]["panoptic_seg"][0].clone()})
    piped = Cfg4Step(seed=3, n_fragments=3, pipeline=True)
    outs = [piped.run() for _ in range(3)]
    assert "panoptic_finish" in outs[-1]                                          # still pending
    assert "panoptic_info" in outs[0] and "panoptic_finish" not in outs[0]        # finished one step later
    piped.flush()
    assert "panoptic_finish" not in outs[-1]
    torch.cuda.synchronize()
    for exp, out in zip(expected, outs):
        assert torch.equal(exp["coords"], out["coords"]) and torch.equal(exp["tsdf"], out["tsdf"])
        assert torch.equal(exp["logits"], out["panoptic_out"][0]["pred_logits"])
        assert torch.equal(exp["masks"], out["panoptic_out"][0]["pred_masks"])
        assert torch.equal(exp["seg"], out["panoptic_info"][0]["panoptic_seg"][0])
