"""GPU parity: HIP back-projection (through the C ABI) vs the CPU oracle and the reference golden
vectors.  Bar: counts / valid set / output order / coords / visibility masks / normalised image
coordinates bit-exact; features within 1e-3 (north_star tolerance; observed ~1e-6)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from eprecon_amd import synthetic as S  # noqa: E402
from oracle import back_project as O  # noqa: E402
from test_oracle_back_project import CASES, ROW_STRIDE, bp_inputs  # noqa: E402

FEAT_TOL = 1e-3


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def hip_run(coords, origin, vs, feats, kr, mv, mode=0, **kw):
    from eprecon_amd import back_project as BP
    return BP.run(_dev(coords), _dev(origin), vs, _dev(feats), _dev(kr), mv, mode, **kw)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "back_project.npz"))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mv", [0, 2])
def test_matches_oracle_and_golden(gold, name, mv):
    window, coords, origin, feats, kr = bp_inputs(gold[name + "_meta"])
    ref = O.back_project(coords, origin, window["voxel_size"], feats, kr, mv, O.MODE_MEAN, want_grid=True)
    got = hip_run(coords, origin, window["voxel_size"], feats, kr, mv, 0, want_grid=True)
    assert got is not None
    assert got["n_valid"] == ref["feats"].shape[0] == int(gold[f"{name}_mv{mv}_nvalid"])
    assert np.array_equal(got["count"].cpu().numpy(), ref["count"])
    assert np.array_equal(got["coords"].cpu().numpy(), ref["coords"])
    assert np.array_equal(got["mask"].cpu().numpy(), ref["mask"])
    assert np.array_equal(got["grid"].cpu().numpy(), ref["grid"])
    f = got["feats"].cpu().numpy()
    assert np.abs(f - ref["feats"]).max() < FEAT_TOL
    # and against the vectors captured from the reference itself
    key = f"{name}_mv{mv}"
    rows = gold[key + "_rows"]
    assert np.array_equal(got["count"].cpu().numpy().astype(np.uint8), gold[key + "_count"])
    assert np.array_equal(got["coords"].cpu().numpy()[rows], gold[key + "_coord_rows"])
    assert np.abs(f[rows] - gold[key + "_feat_rows"]).max() < FEAT_TOL
    rowsum = f.sum(axis=1, dtype=np.float64).astype(np.float32)[::ROW_STRIDE]
    np.testing.assert_allclose(rowsum, gold[key + "_rowsum"], atol=FEAT_TOL, rtol=0)


def test_channels_last_input_gives_identical_result(gold):
    window, coords, origin, feats, kr = bp_inputs(gold["cfg2_l1_meta"])
    from eprecon_amd import back_project as BP
    a = hip_run(coords, origin, 0.04, feats, kr, 2)
    f_cl = BP.to_channels_last(_dev(feats))
    assert f_cl.shape == feats.shape and f_cl.stride()[2] == 1
    b = BP.run(_dev(coords), _dev(origin), 0.04, f_cl, _dev(kr), 2)
    assert torch.equal(a["feats"], b["feats"]) and torch.equal(a["coords"], b["coords"])
    # the re-layout kernel itself is a pure permutation
    assert torch.equal(f_cl.contiguous(), _dev(feats))


@pytest.mark.parametrize("name", ["cfg1_l0", "cfg2_l1"])
def test_tile_order_switch_gives_identical_result(gold, name, monkeypatch):
    """EPRECON_BP_XCD_SLABS=0 (tiles in hardware block order instead of one contiguous slab of the raster per XCD: the A/B of
    DESIGN.md 3a's fetch amplification) changes where a tile runs, not what it computes"""
    window, coords, origin, feats, kr = bp_inputs(gold[name + "_meta"])
    a = hip_run(coords, origin, window["voxel_size"], feats, kr, 2, 0)
    monkeypatch.setenv("EPRECON_BP_XCD_SLABS", "0")
    b = hip_run(coords, origin, window["voxel_size"], feats, kr, 2, 0)
    assert a["n_valid"] == b["n_valid"] and torch.equal(a["coords"], b["coords"]) and torch.equal(a["feats"], b["feats"])
    assert torch.equal(a["count"], b["count"])


@pytest.mark.parametrize("name", ["cfg1_l0", "cfg1b2_l1"])
def test_depth_channel(gold, name):
    from eprecon_amd.back_project import back_project
    window, coords, origin, feats, kr = bp_inputs(gold[name + "_meta"])
    ref = O.back_project(coords, origin, 0.04, feats, kr, 2, O.MODE_MEAN_DEPTH)
    out = back_project(_dev(coords).float(), _dev(origin), 0.04, _dev(feats), _dev(kr), 2)
    f = out[0].cpu().numpy()
    assert f.shape == ref["feats"].shape
    assert out[1].dtype == torch.float32 and np.array_equal(out[1].cpu().numpy(), ref["coords"].astype(np.float32))
    assert np.abs(f - ref["feats"]).max() < FEAT_TOL
    rows = gold[f"{name}_mv2_rows"]
    assert np.abs(f[rows, -1] - gold[f"{name}_mv2_depth_rows"]).max() < FEAT_TOL


@pytest.mark.parametrize("name", ["cfg1_l0", "cfg2_l1"])
def test_variance_mode(gold, name):
    window, coords, origin, feats, kr = bp_inputs(gold[name + "_meta"])
    ref = O.back_project(coords, origin, 0.04, feats, kr, 2, O.MODE_VARIANCE)
    got = hip_run(coords, origin, 0.04, feats, kr, 2, 2, want_mean=True)
    assert np.array_equal(got["coords"].cpu().numpy(), ref["coords"])
    assert np.abs(got["feats"].cpu().numpy() - ref["feats"]).max() < FEAT_TOL
    assert np.abs(got["mean"].cpu().numpy() - ref["mean"]).max() < FEAT_TOL


def test_module_api_and_none_convention():
    from eprecon_amd.back_project import Back_Project
    window = S.make_window(seed=3, width=320, height=240, n_vox=(32, 32, 32))
    c, h, w = S.pyramid_shapes(240, 320)[0]
    feats = S.make_features(5, 9, (c, h, w))
    coords = S.dense_coords((32, 32, 32), 4)
    kr = np.ascontiguousarray(window["proj_matrices"][:, 0][:, None])
    origin = window["vol_origin_partial"][None].copy()
    mod = Back_Project(c).cuda()
    out = mod(_dev(coords), _dev(origin), 0.04, _dev(feats), _dev(kr), 2)
    assert len(out) == 5 and out[2] is None and out[3] is None
    assert out[0].shape[1] == c and out[1].dtype == torch.int32 and out[4].shape[0] == coords.shape[0]
    far = origin.copy()
    far[0, 1] -= 50.0
    assert mod(_dev(coords), _dev(far), 0.04, _dev(feats), _dev(kr), 1) is None
    out0 = mod(_dev(coords), _dev(far), 0.04, _dev(feats), _dev(kr), 0)
    assert out0[0].shape[0] == coords.shape[0] and not out0[0].any() and not out0[4].any()
    # the reference defines Back_Project and get_img_feats in models/occupancy_initialization.py (:185, :264) and imports them
    # from there (models/neucon_network.py:20): the mirror module exports the same names
    from eprecon_amd.occupancy_initialization import Back_Project as BP2, Occupancy_Initialization, get_img_feats  # noqa: F401
    assert BP2 is Back_Project
    f = get_img_feats(_dev(coords), _dev(origin), 0.04, _dev(feats), _dev(kr), 2)
    assert torch.equal(f, out[0])
    assert get_img_feats(_dev(coords), _dev(far), 0.04, _dev(feats), _dev(kr), 1).shape == (0, c)


def test_ragged_sparse_list_and_odd_channels():
    """non-dense, non-multiple-of-block voxel list; C not a multiple of 4 (scalar path)"""
    rng = np.random.default_rng(11)
    window = S.make_window(seed=4)
    coords = S.dense_coords((96, 96, 96), 2)
    keep = np.sort(rng.choice(coords.shape[0], size=70001, replace=False))
    coords = np.ascontiguousarray(coords[keep])
    for ch in (7, 12):
        feats = S.make_features(9, 9, (ch, 60, 80))
        kr = np.ascontiguousarray(window["proj_matrices"][:, 1][:, None])
        origin = window["vol_origin_partial"][None]
        ref = O.back_project(coords, origin, 0.04, feats, kr, 3)
        got = hip_run(coords, origin, 0.04, feats, kr, 3)
        assert np.array_equal(got["coords"].cpu().numpy(), ref["coords"])
        assert np.array_equal(got["count"].cpu().numpy(), ref["count"])
        assert np.abs(got["feats"].cpu().numpy() - ref["feats"]).max() < FEAT_TOL


def test_empty_input_list():
    window = S.make_window(seed=4)
    feats = S.make_features(9, 9, (24, 30, 40))
    kr = np.ascontiguousarray(window["proj_matrices"][:, 2][:, None])
    got = hip_run(np.zeros((0, 4), np.int32), window["vol_origin_partial"][None], 0.04, feats, kr, 0)
    assert got is None  # the reference returns None when a batch element has no valid voxel


def test_linearity_and_idempotence_at_full_size():
    """size-independent properties at the BASELINE size (dense 96^3, C=24, 120x160):
    back-projection is linear in the feature maps, and a second run is bit-identical."""
    from eprecon_amd import back_project as BP
    window = S.make_window(seed=0)
    c, h, w = S.pyramid_shapes()[0]
    fa = _dev(S.make_features(21, 9, (c, h, w)))
    fb = _dev(S.make_features(22, 9, (c, h, w)))
    coords = _dev(S.dense_coords((96, 96, 96), 1))
    kr = _dev(window["proj_matrices"][:, 0][:, None])
    origin = _dev(window["vol_origin_partial"][None])
    ra = BP.run(coords, origin, 0.04, fa, kr, 2)
    rb = BP.run(coords, origin, 0.04, fb, kr, 2)
    rab = BP.run(coords, origin, 0.04, 2.0 * fa - fb, kr, 2)
    ra2 = BP.run(coords, origin, 0.04, fa, kr, 2)
    assert torch.equal(ra["feats"], ra2["feats"]) and torch.equal(ra["coords"], ra2["coords"])
    assert torch.equal(ra["coords"], rab["coords"]) and torch.equal(ra["count"], rb["count"])
    assert (rab["feats"] - (2.0 * ra["feats"] - rb["feats"])).abs().max().item() < 1e-4
    # compaction is stable: output rows are the input rows with count >= 2, in input order
    keep = ra["count"] >= 2
    assert torch.equal(ra["coords"], coords[keep])


def test_async_queue_matches_blocking_run():
    """three levels queued back to back (run_async) give the bits of the blocking entry, and the None
    convention survives the deferred read-back"""
    from eprecon_amd import back_project as BP
    window = S.make_window(seed=6)
    shapes = S.pyramid_shapes(480, 640)
    origin = _dev(window["vol_origin_partial"][None].copy())
    args = []
    for lvl, interval in ((2, 4), (1, 2), (0, 1)):
        feats = _dev(S.make_features(40 + lvl, 9, shapes[lvl]))
        kr = _dev(np.ascontiguousarray(window["proj_matrices"][:, lvl][:, None]))
        args.append((_dev(S.dense_coords((96, 96, 96), interval)), origin, 0.04, feats, kr, 3))
    pend = [BP.run_async(*a) for a in args]
    far = origin.clone()
    far[0, 1] -= 50.0
    pend_none = BP.run_async(args[0][0], far, 0.04, args[0][3], args[0][4], 1)
    for a, p in zip(args, pend):
        ref, got = BP.run(*a), p.result()
        assert got["n_valid"] == ref["n_valid"] and got["n_valid_per_batch"] == ref["n_valid_per_batch"]
        for key in ("feats", "coords", "count"):
            assert torch.equal(got[key], ref[key])
    assert pend_none.result() is None


def test_more_than_32_views_is_rejected_loudly():
    """the visible-view bitmask is 32 bits wide: 33 views must raise, not wrap"""
    from eprecon_amd import _lib
    window = S.make_window(seed=4)
    feats = np.zeros((33, 1, 8, 30, 40), np.float32)
    kr = np.repeat(window["proj_matrices"][:1, 2][:, None], 33, axis=0)
    coords = S.dense_coords((96, 96, 96), 4)[:1000]
    with pytest.raises(_lib.EpreconError):
        hip_run(coords, window["vol_origin_partial"][None], 0.04, feats, np.ascontiguousarray(kr), 1)
