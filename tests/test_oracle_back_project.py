"""Pins oracle/c/back_project_oracle.c against golden vectors captured from the reference's own
Back_Project.forward / ops.back_project (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from eprecon_amd import synthetic as S
from oracle import back_project as O

ROW_STRIDE = 32


def bp_inputs(meta):
    lvl, interval, batch, seed, width, height, nvox, fseed = [int(x) for x in meta]
    window = S.make_window(seed=seed, width=width, height=height, n_vox=(nvox,) * 3)
    c, h, w = S.pyramid_shapes(height, width)[lvl]
    feats = S.make_features(fseed, 9, (c, h, w), batch=batch)
    coords = S.dense_coords(window["n_vox"], interval, batch=batch)
    kr = np.ascontiguousarray(np.repeat(window["proj_matrices"][:, lvl][:, None], batch, axis=1))
    origin = np.repeat(window["vol_origin_partial"][None], batch, axis=0).copy()
    if batch > 1:
        origin[1:, 0] += 0.36
    return window, coords, origin, feats, kr


CASES = ["cfg1_l0", "cfg1b2_l1", "cfg2_l2", "cfg2_l1", "cfg2_l0"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "back_project.npz"))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mv", [0, 2])
def test_oracle_matches_reference_golden(gold, name, mv):
    window, coords, origin, feats, kr = bp_inputs(gold[name + "_meta"])
    res = O.back_project(coords, origin, window["voxel_size"], feats, kr, mv, O.MODE_MEAN,
                         want_grid=True)
    key = f"{name}_mv{mv}"
    # integer / index results: bit exact
    assert np.array_equal(res["count"].astype(np.uint8), gold[key + "_count"])
    assert res["feats"].shape[0] == int(gold[key + "_nvalid"])
    rows = gold[key + "_rows"]
    assert np.array_equal(res["coords"][rows], gold[key + "_coord_rows"])
    assert np.array_equal(res["mask"][:, rows], gold[key + "_mask_rows"])
    # the normalised image coordinates are bit exact too (same fma chain as torch's CPU bmm)
    assert np.array_equal(res["grid"][:, rows], gold[key + "_grid_rows"])
    # features: fp32 tolerance 1e-3 per north_star; observed ~5e-7
    np.testing.assert_allclose(res["feats"][rows], gold[key + "_feat_rows"], atol=1e-5, rtol=0)
    rowsum = res["feats"].sum(axis=1, dtype=np.float64).astype(np.float32)[::ROW_STRIDE]
    np.testing.assert_allclose(rowsum, gold[key + "_rowsum"], atol=2e-4, rtol=0)


@pytest.mark.parametrize("name", ["cfg1_l0", "cfg1b2_l1"])
def test_oracle_depth_channel(gold, name):
    window, coords, origin, feats, kr = bp_inputs(gold[name + "_meta"])
    res = O.back_project(coords, origin, window["voxel_size"], feats, kr, 2, O.MODE_MEAN_DEPTH)
    rows = gold[f"{name}_mv2_rows"]
    assert res["feats"].shape[1] == feats.shape[2] + 1
    np.testing.assert_allclose(res["feats"][rows, -1], gold[f"{name}_mv2_depth_rows"], atol=1e-5)


def test_oracle_returns_none_when_nothing_visible():
    window = S.make_window(seed=3, width=320, height=240, n_vox=(32, 32, 32))
    c, h, w = S.pyramid_shapes(240, 320)[0]
    feats = S.make_features(5, 9, (c, h, w))
    coords = S.dense_coords((32, 32, 32), 4)
    kr = np.ascontiguousarray(window["proj_matrices"][:, 0][:, None])
    origin = window["vol_origin_partial"][None].copy()
    origin[0, 1] -= 50.0  # far behind every camera
    assert O.back_project(coords, origin, 0.04, feats, kr, 1) is None
    # min_view = 0 keeps every voxel, with all-zero features
    res = O.back_project(coords, origin, 0.04, feats, kr, 0)
    assert res["feats"].shape[0] == coords.shape[0] and not res["feats"].any()
    assert not res["count"].any()
