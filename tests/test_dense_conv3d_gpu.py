"""Dense-grid forms of the 3x3x3 stride-1 convolution (conv3d_tile16_kernel / conv3d_tile_narrow_kernel, csrc/sparse_conv.hip)
against (a) the gather-GEMM form of the same layer through the [27, N] kernel map — equal within fp32 round-off: the tile
kernels multiply four input channels per MFMA, another summation order — and (b) the numpy oracle (oracle/sparse.py, restating
the spconv SubMConv3d semantics of models/modules.py:249-271) within 1e-3."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import sparse as OS  # noqa: E402

TOL = 1e-3


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _lib_last_kernel():
    from eprecon_amd import _lib
    return _lib.last_conv_kernel()


def grid_set(rng, dims, stride, fill, batch=0):
    """raster-ordered (x slowest, z fastest) subset of the dense grid: int32[N,4] (b, x, y, z) in finest-voxel units"""
    gx, gy, gz = dims
    x, y, z = np.meshgrid(np.arange(gx), np.arange(gy), np.arange(gz), indexing="ij")
    c = np.stack([np.full(x.size, batch), x.ravel() * stride, y.ravel() * stride, z.ravel() * stride], 1).astype(np.int32)
    keep = rng.random(len(c)) < fill
    if fill < 1.0:   # carve a corner out as well: whole tiles without a voxel
        keep &= ~((c[:, 1] < 5 * stride) & (c[:, 2] < 5 * stride) & (c[:, 3] < 9 * stride))
    return c[keep]


def sets(dims, stride, fill, seed):
    from eprecon_amd.sparse import DenseMap, VoxelSet
    rng = np.random.default_rng(seed)
    c = grid_set(rng, dims, stride, fill)
    vs = VoxelSet(dev(c), stride, dims=dims)
    dm = DenseMap(vs, dims)
    assert dm.off_grid() == 0
    return rng, c, vs, dm


def test_rank_volume_matches_definition():
    rng, c, vs, dm = sets((13, 9, 21), 2, 0.7, 1)
    rank = dm.rank[:-1].cpu().numpy().reshape(13, 9, 21)
    ref = np.full((13, 9, 21), -1, np.int32)
    ref[c[:, 1] // 2, c[:, 2] // 2, c[:, 3] // 2] = np.arange(len(c))
    assert np.array_equal(rank, ref)


def test_rank_volume_counts_voxels_off_the_grid():
    from eprecon_amd.sparse import DenseMap, VoxelSet
    c = np.array([[0, 0, 0, 0], [0, 2, 2, 2], [0, 3, 2, 2], [0, 40, 0, 0], [0, -2, 0, 0]], np.int32)  # odd, outside, negative
    from eprecon_amd import _lib
    _lib.take_deferred(None)
    assert DenseMap(VoxelSet(dev(c), 2), (8, 8, 8)).off_grid() == 3
    # production: nobody calls off_grid() — the map deferred its own check, and the NEXT blocking count read on the stream raises
    # instead of leaving three output rows of every convolution on the set unwritten
    with pytest.raises(_lib.EpreconError, match="not on the grid"):
        _lib.read_counts(torch.zeros(1, dtype=torch.int32, device="cuda"))
    assert _lib.take_deferred(None) == []


def test_pending_batchnorm_on_load_and_column_slices():
    """the producer's BatchNorm (+ReLU) applied while the halo is staged; input / output as slices of wider buffers"""
    from eprecon_amd import sparse as SP
    rng, c, vs, dm = sets((20, 20, 20), 2, 0.8, 7)
    n = len(c)
    wide_in = dev(rng.standard_normal((n, 64)).astype(np.float32))
    x = wide_in[:, 16:48]                      # 32 channels at a 64-float pitch, 16-byte aligned
    w = dev((rng.standard_normal((27, 32, 16)) / 30).astype(np.float32))
    scale, shift = dev(rng.random(32).astype(np.float32) + 0.5), dev(rng.standard_normal(32).astype(np.float32))
    out_a = torch.zeros((n, 48), device="cuda")
    out_b = torch.zeros((n, 48), device="cuda")
    _, pa = SP.conv_stats(x, w, vs.kernel_map(3), in_affine=(scale, shift, True), out=out_a[:, 16:32])
    _, pb = SP.conv_stats(x, w, dm, in_affine=(scale, shift, True), out=out_b[:, 16:32])
    assert float((out_a - out_b).abs().max()) < 2e-5
    xn = np.maximum(x.cpu().numpy() * scale.cpu().numpy() + shift.cpu().numpy(), 0)
    ref = OS.sparse_conv(xn, OS.kernel_map(c, c, 3, 2), w.cpu().numpy(), None)
    assert np.abs(out_b[:, 16:32].cpu().numpy() - ref).max() < TOL
    assert float(out_b[:, :16].abs().max()) == 0 and float(out_b[:, 32:].abs().max()) == 0


@pytest.mark.parametrize("dims,stride,fill,cin", [((48, 48, 48), 2, 0.85, 32), ((13, 9, 21), 1, 0.6, 16), ((16, 16, 16), 1, 1.0, 24)])
def test_single_column_kernel(dims, stride, fill, cin):
    """C_out == 1 (the occupancy-logit layer): plain fma chains, other summation order than the MFMA form -> tolerance"""
    from eprecon_amd import sparse as SP
    rng, c, vs, dm = sets(dims, stride, fill, cin)
    n = len(c)
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((27, cin, 1)) / np.sqrt(27 * cin)).astype(np.float32)
    b = rng.standard_normal(1).astype(np.float32)
    y, part = SP.sparse_conv_fused(dev(x), dev(w), dm, dev(b), bn_partial=True)
    ref = OS.sparse_conv(x, OS.kernel_map(c, c, 3, stride), w, b)
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-4
    # the summaries are those of the stored values
    p = part.cpu().numpy().astype(np.float64)
    cnt = p[:, 0, 0]
    assert cnt.sum() == n
    mean = (cnt * p[:, 1, 0]).sum() / n
    m2 = (p[:, 2, 0] + cnt * (p[:, 1, 0] - mean) ** 2).sum()
    yv = y.cpu().numpy().astype(np.float64)[:, 0]
    assert abs(mean - yv.mean()) < 1e-5 and abs(m2 / n - yv.var()) < 1e-5 * max(1.0, yv.var())


@pytest.mark.parametrize("dims,stride,fill,cin,cout", [
    ((48, 48, 48), 2, 0.85, 32, 32),    # the initialisation stack's layers
    ((48, 48, 48), 2, 0.85, 32, 16),
    ((48, 48, 48), 2, 0.85, 16, 16),
    ((13, 9, 21), 2, 0.6, 16, 24),      # ragged grid, partial column tile
    ((20, 12, 16), 1, 0.5, 48, 32),
    ((16, 16, 16), 1, 0.9, 64, 8),
])
def test_16_row_tile_kernel(monkeypatch, dims, stride, fill, cin, cout):
    """conv3d_tile16_kernel (v_mfma_f32_16x16x4_f32, 2x4x8-cell tiles): another summation order than the 32x32x2 kernels
    (four input channels per MFMA) -> equal to the gather form within fp32 round-off, and to the oracle within 1e-3; every
    epilogue it implements: bias + ReLU + residual, row-wise LayerNorm, BatchNorm summaries, pending BatchNorm on load"""
    from eprecon_amd import sparse as SP
    monkeypatch.delenv("EPRECON_CONV_DENSE3D", raising=False)
    rng, c, vs, dm = sets(dims, stride, fill, cin * 77 + cout)
    n = len(c)
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((n, cout)).astype(np.float32)
    dx, dw, db, dres = dev(x), dev(w), dev(b), dev(res)
    nbr = vs.kernel_map(3)
    assert dm.kind(dx, cin, cout) == 2
    y_map, _ = SP.sparse_conv_fused(dx, dw, nbr, db, relu=True, residual=dres)
    y_16, _ = SP.sparse_conv_fused(dx, dw, dm, db, relu=True, residual=dres)
    assert float((y_map - y_16).abs().max()) < 2e-5
    ref = OS.sparse_conv(x, OS.kernel_map(c, c, 3, stride), w, b)
    assert np.abs(y_16.cpu().numpy() - (np.maximum(ref, 0) + res)).max() < TOL
    # BatchNorm summaries + pending BatchNorm (+ReLU) of the producer on load
    scale, shift = dev(rng.random(cin).astype(np.float32) + 0.5), dev(rng.standard_normal(cin).astype(np.float32))
    ya, pa = SP.conv_stats(dx, dw, nbr, in_affine=(scale, shift, True))
    yb, pb = SP.conv_stats(dx, dw, dm, in_affine=(scale, shift, True))
    assert float((ya - yb).abs().max()) < 2e-5 and float(pb[:, 0, 0].sum()) == n
    g, z = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    sa, ta = SP.bn_affine(pa, g, z, 1e-5)
    sb, tb = SP.bn_affine(pb, g, z, 1e-5)
    assert torch.allclose(sa, sb, rtol=1e-4, atol=1e-6) and torch.allclose(ta, tb, rtol=1e-4, atol=1e-4)
    # row-wise LayerNorm epilogue
    lg, lb = dev(rng.standard_normal(cout).astype(np.float32)), dev(rng.standard_normal(cout).astype(np.float32))
    r_ = dres if cin == cout else None
    a = SP.sparse_conv_ln(dx, dw, nbr, db, lg, lb, 1e-5, relu=True, residual=r_, post_relu=(cout == 16))
    d = SP.sparse_conv_ln(dx, dw, dm, db, lg, lb, 1e-5, relu=True, residual=r_, post_relu=(cout == 16))
    assert float((a - d).abs().max()) < 1e-4
    # output as a column slice of a wider buffer
    wide = torch.zeros((n, cout + 8), device="cuda")
    SP.sparse_conv(dx, dw, dm, db, out=wide[:, 4:4 + cout])
    assert float((wide[:, 4:4 + cout] - SP.sparse_conv(dx, dw, nbr, db)).abs().max()) < 2e-5 and float(wide[:, :4].abs().max()) == 0


def test_levels_select_the_kernels(monkeypatch):
    from eprecon_amd import sparse as SP
    rng, c, vs, dm = sets((16, 16, 16), 1, 1.0, 11)
    x = dev(rng.standard_normal((len(c), 32)).astype(np.float32))
    x40 = dev(rng.standard_normal((len(c), 40)).astype(np.float32))
    monkeypatch.setenv("EPRECON_CONV_DENSE3D", "1")
    assert dm.kind(x, 32, 1) == 1 and dm.kind(x, 32, 32) == 0
    w = dev((rng.standard_normal((27, 32, 32)) / 30).astype(np.float32))
    assert torch.equal(SP.sparse_conv(x, w, dm), SP.sparse_conv(x, w, vs.kernel_map(3)))   # falls back to the map
    monkeypatch.delenv("EPRECON_CONV_DENSE3D")                                            # the default: level 2
    assert dm.kind(x, 32, 1) == 1 and dm.kind(x, 32, 32) == 2 and dm.kind(x, 32, 16) == 2
    assert dm.kind(x40, 40, 32) == 0 and dm.kind(x, 32, 48) == 0                          # not 16-row shapes: kernel map
    w48 = dev((rng.standard_normal((27, 32, 48)) / 30).astype(np.float32))
    assert torch.equal(SP.sparse_conv(x, w48, dm), SP.sparse_conv(x, w48, vs.kernel_map(3)))
    assert _lib_last_kernel() != "conv3d_tile16_kernel"
    SP.sparse_conv(x, w, dm)
    assert _lib_last_kernel() == "conv3d_tile16_kernel"


def test_sparse_set_falls_back_to_the_kernel_map():
    """a thinly filled grid keeps the gather form (VoxelSet.conv_map), and shapes the tile kernel does not take fall back"""
    from eprecon_amd import sparse as SP
    from eprecon_amd.sparse import DenseMap, VoxelSet
    rng = np.random.default_rng(3)
    c = grid_set(rng, (24, 24, 24), 1, 0.1)
    vs = VoxelSet(dev(c), 1, dims=(24, 24, 24))
    assert torch.is_tensor(vs.conv_map(3))
    full = VoxelSet(dev(grid_set(rng, (12, 12, 12), 1, 1.0)), 1, dims=(12, 12, 12))
    dm = full.conv_map(3)
    assert isinstance(dm, DenseMap)
    x = dev(rng.standard_normal((full.n, 80)).astype(np.float32))       # 80 input channels: not a tile-kernel shape
    w = dev((rng.standard_normal((27, 80, 8)) / 40).astype(np.float32))
    assert not dm.takes(x, 80, 8)
    assert torch.equal(SP.sparse_conv(x, w, dm), SP.sparse_conv(x, w, full.kernel_map(3)))
