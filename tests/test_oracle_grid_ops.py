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


def test_dense_coords_is_built_once_per_shape_and_matches_generate_grid():
    """eprecon_amd.generate_grids.dense_coords hands out one cached raster per (volume, interval, batch size, device): same
    object on the second call, rows = [b, x, y, z] of generate_grid's x-major raster (models/neucon_network.py:246-251)"""
    import torch
    from eprecon_amd.generate_grids import dense_coords, generate_grid
    a, dims = dense_coords((24, 16, 8), 4, 2, device=torch.device("cpu"))
    b, dims_b = dense_coords((24, 16, 8), 4, 2, device=torch.device("cpu"))
    assert a is b and dims == dims_b == (6, 4, 2)
    grid, _ = generate_grid((24, 16, 8), 4, device=torch.device("cpu"))
    n = grid.shape[1]
    assert a.shape == (2 * n, 4) and a.dtype == torch.int32
    for batch in range(2):
        rows = a[batch * n:(batch + 1) * n]
        assert bool((rows[:, 0] == batch).all()) and torch.equal(rows[:, 1:], grid.t().to(torch.int32))
    other, _ = dense_coords((24, 16, 8), 2, 2, device=torch.device("cpu"))
    assert other is not a and other.shape[0] == 2 * 12 * 8 * 4
