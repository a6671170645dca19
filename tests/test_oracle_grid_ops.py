"""CPU: oracle/grid_ops.py pinned against golden vectors from the reference's generate_grid,
NeuConNet.upsample and the erode/dilate selection (tests/golden/make_golden.py: grid_ops)."""
import os

import numpy as np
import pytest

from oracle import grid_ops as OG


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "grid_ops.npz"))


def selection_inputs(gold):
    valid = gold["sel_valid"]
    xyz = np.argwhere(valid) * 2           # finest-voxel units of the interval-2 grid, raster order
    coords = np.concatenate([np.zeros((len(xyz), 1), int), xyz], 1).astype(np.int32)
    return gold["sel_logit_vol"][valid].astype(np.float32), coords


def test_generate_grid(gold):
    for interval in (1, 2, 4):
        g, dims = OG.generate_grid([96, 96, 96], interval)
        assert tuple(gold[f"grid_i{interval}_dims"]) == dims
        assert np.array_equal(g[:, :200], gold[f"grid_i{interval}_head"])
        chk = np.array([g.astype(np.float64).sum(), (g.astype(np.float64) * np.arange(1, g.shape[1] + 1)).sum()])
        assert np.allclose(chk, gold[f"grid_i{interval}_checksum"], rtol=0, atol=0)


def test_synthetic_dense_coords_match_reference_grid(gold):
    from eprecon_amd import synthetic as S
    c = S.dense_coords((96, 96, 96), 4)
    assert np.array_equal(c[:200, 1:].T.astype(np.float32), gold["grid_i4_head"])


def test_upsample(gold):
    rng = np.random.default_rng(31)
    coords = rng.integers(0, 24, size=(500, 3)) * 4
    coords = np.concatenate([rng.integers(0, 2, size=(500, 1)), coords], 1).astype(np.int32)
    feat = rng.standard_normal((500, 7)).astype(np.float32)
    uf, uc = OG.upsample(feat, coords, 2)
    assert np.array_equal(uf, gold["up_feat"]) and np.array_equal(uc, gold["up_coords"])


def test_init_selection(gold):
    logit, coords = selection_inputs(gold)
    got = OG.init_select(logit, coords, 1)
    assert len(got) > 100
    assert np.array_equal(got[:, 1:], gold["sel_coords"]) and not got[:, 0].any()
