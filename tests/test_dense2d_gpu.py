"""The 2D fusion blocks of the initialisation branch on the HIP gather-GEMM path (dense2d.py) against
the PyTorch modules they mirror (models/modules.py:313-399 of the reference; the modules themselves are
pinned to the reference's state_dict / outputs by tests/test_dense_blocks.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3  # north_star: fp32 features within 1e-3 (observed ~1e-5)


def _dev():
    return torch.device("cuda:0")


def _cl(x):
    return x.contiguous(memory_format=torch.channels_last)


def test_pixel_map_matches_unfold():
    from eprecon_amd import dense2d as D2
    v, h, w = 3, 5, 7
    g = D2.PixelGrid(v, h, w, _dev())
    nbr = g.kernel_map(3).cpu()
    idx = torch.arange(v * h * w, dtype=torch.float32).view(v, 1, h, w) + 1.0  # 0 = padding
    ref = F.unfold(idx, 3, padding=1).permute(1, 0, 2).reshape(9, -1).to(torch.int64) - 1
    assert torch.equal(nbr.to(torch.int64), ref)


@pytest.mark.parametrize("v,h,w,ci,co,k", [(9, 30, 40, 80, 80, 3), (9, 30, 40, 320, 80, 1), (2, 17, 23, 24, 12, 3),
                                           (9, 60, 80, 144, 32, 1), (1, 8, 8, 12, 12, 3), (3, 16, 16, 40, 200, 3),
                                           (9, 60, 80, 16, 200, 1)])
def test_fused_conv_epilogues(v, h, w, ci, co, k):
    """bias + ReLU + residual epilogue, column split for short lists, and the BatchNorm summaries"""
    from eprecon_amd import dense2d as D2, sparse as SP
    torch.manual_seed(v * 1000 + ci + co + k)
    dev = _dev()
    x = _cl(torch.randn(v, ci, h, w, device=dev))
    conv = torch.nn.Conv2d(ci, co, k, padding="same").to(dev)
    res = torch.randn(v * h * w, co, device=dev)
    g = D2.PixelGrid(v, h, w, dev)
    wk = D2.packed_weight(conv)
    with torch.no_grad():
        ref = F.relu(conv(x)).permute(0, 2, 3, 1).reshape(v * h * w, co) + res
        out, partial = SP.sparse_conv_fused(D2.rows_of(x), wk if k > 1 else wk[0], g.kernel_map(k), conv.bias,
                                            relu=True, residual=res, bn_partial=True)
    assert (out - ref).abs().max().item() < TOL
    # summaries: counts add up to the row count, the merged statistics are the column mean / variance
    n = v * h * w
    assert torch.allclose(partial[:, 0, :].sum(0), torch.full((co,), float(n), device=dev))
    y = SP.batchnorm_apply_partials(out, partial, relu=False)
    ref_bn = F.batch_norm(out, None, None, training=True, eps=1e-5)
    assert (y - ref_bn).abs().max().item() < TOL
    # bit-identical to the stand-alone statistics pass?  Not required (different block sizes), but both are
    # deterministic: a second run reproduces the bits
    out2, partial2 = SP.sparse_conv_fused(D2.rows_of(x), wk if k > 1 else wk[0], g.kernel_map(k), conv.bias,
                                          relu=True, residual=res, bn_partial=True)
    assert torch.equal(out, out2) and torch.equal(partial, partial2)


@pytest.fixture(autouse=True, params=["finalize", "acc"])
def bn_mode(request):
    """every test of the rows path runs twice: with a finalize launch behind every layer (rounds 1-5) and with the layers'
    BatchNorms left in order-independent accumulator blocks that their consumers finish (round 6: BatchNorm form (c))"""
    from eprecon_amd import dense2d as D2
    if request.param == "finalize":
        old, D2.BN_ACC = D2.BN_ACC, False
        try:
            yield request.param
        finally:
            D2.BN_ACC = old
    else:
        with D2.bn_pass(D2.BnArena(_dev())):
            yield request.param


def test_accumulator_blocks_give_the_finalize_launch_result_and_the_same_bits_every_run():
    """form (c) against the Chan-merged summaries of the same launch: (scale, shift) of one 3x3 layer on 43,200 pixel rows
    agree to fp32 round-off; two runs of the whole accumulating 2D stack are bit-identical (integer atomics commute)"""
    import ctypes
    from eprecon_amd import _lib, dense2d as D2, sparse as SP
    from eprecon_amd.modules import Conv2d_Block
    from eprecon_amd.occupancy_initialization import Occupancy_Initialization
    dev = _dev()
    torch.manual_seed(5)
    blk = Conv2d_Block(32, 32, 3).to(dev).train()
    blk.bn.weight.data.uniform_(0.5, 1.5)
    blk.bn.bias.data.uniform_(-1, 1)
    x = torch.randn(9, 32, 60, 80, device=dev) * 3 + 7        # a mean well away from zero
    g = D2.PixelGrid.get(9, 60, 80, dev)
    with torch.no_grad():
        with D2.bn_pass(D2.BnArena(dev)):
            a = blk.run_act(D2.Act(D2.rows_of(_cl(x))), g)
            assert a.acc is not None and a.scale is None
            sc, sh = a.affine()
        with D2.bn_pass(None):
            b = blk.run_act(D2.Act(D2.rows_of(_cl(x))), g)
        assert torch.equal(a.rows, b.rows)
        assert (sc - b.scale).abs().max().item() < 1e-5 * b.scale.abs().max().item() + 1e-7
        assert (sh - b.shift).abs().max().item() < 1e-5 * b.shift.abs().max().item() + 1e-6
        ref = blk(x)
        assert (D2.maps_of(D2.materialize(a), 9, 60, 80) - ref).abs().max().item() < TOL
        net = Occupancy_Initialization([80, 40, 24], 32, 9).to(dev).train()
        net.use_hip_graph = False
        f = [torch.randn(9, 80, 15, 20, device=dev), torch.randn(9, 40, 30, 40, device=dev), torch.randn(9, 24, 60, 80, device=dev)]
        old, D2.BN_ACC = D2.BN_ACC, True
        try:
            r1 = net.feat_fusion_pre(*f).clone()
            r2 = net.feat_fusion_pre(*f).clone()
        finally:
            D2.BN_ACC = old
        assert torch.equal(r1, r2)


@pytest.mark.parametrize("c,h,w", [(24, 30, 40), (40, 15, 20), (80, 8, 10)])
def test_fusion_block_rows(c, h, w):
    from eprecon_amd import dense2d as D2
    from eprecon_amd.modules import Fusion_Block
    torch.manual_seed(c)
    dev = _dev()
    blk = Fusion_Block(c).to(dev).train()
    for p in blk.parameters():  # non-trivial BatchNorm affine parameters
        if p.dim() == 1:
            p.data.uniform_(0.5, 1.5)
    x = torch.randn(9, c, h, w, device=dev)
    with torch.no_grad():
        ref = blk(x)  # PyTorch-ROCm modules (NCHW)
        g = D2.PixelGrid.get(9, h, w, dev)
        out = blk.run_rows(D2.rows_of(_cl(x)), g)
    got = D2.maps_of(out, 9, h, w)
    assert (got - ref).abs().max().item() < TOL


def test_residual_block_rows():
    from eprecon_amd import dense2d as D2
    from eprecon_amd.modules import Conv2d_Residual_Block, Conv2d_Block
    torch.manual_seed(3)
    dev = _dev()
    x = torch.randn(9, 32, 60, 80, device=dev)
    g = D2.PixelGrid.get(9, 60, 80, dev)
    for blk in (Conv2d_Residual_Block(32, 3).to(dev).train(), Conv2d_Block(32, 16, 3).to(dev).train(),
                Conv2d_Block(32, 48, 1).to(dev).train()):
        with torch.no_grad():
            ref = blk(x)
            out = blk.run_rows(D2.rows_of(_cl(x)), g)
        assert (D2.maps_of(out, 9, 60, 80) - ref).abs().max().item() < TOL


def test_feat_fusion_rows_matches_modules(monkeypatch):
    """whole 2D stack: HIP rows path vs the PyTorch-ROCm module path on the same weights"""
    from eprecon_amd.occupancy_initialization import Occupancy_Initialization
    torch.manual_seed(0)
    dev = _dev()
    net = Occupancy_Initialization([80, 40, 24], 32, 9).to(dev).train()
    f1 = torch.randn(9, 80, 15, 20, device=dev)
    f2 = torch.randn(9, 40, 30, 40, device=dev)
    f4 = torch.randn(9, 24, 60, 80, device=dev)
    with torch.no_grad():
        got = net.feat_fusion_pre(f1, f2, f4)
        net.use_hip_conv = False
        ref = net.feat_fusion_pre(f1, f2, f4)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL


@pytest.mark.parametrize("cin,cmid,cout", [(12, 12, 12), (24, 12, 20), (40, 40, 40), (20, 36, 8), (32, 32, 32)])
def test_tile_kernel_chain_with_ragged_tiles(cin, cmid, cout):
    """image-tile kernel (3x3, narrow layers, >= 256 tiles): 50 x 70 images leave partial tiles in both
    directions; two chained layers exercise the pending-BatchNorm-on-load prologue, the second one with a
    residual; C_in = 40 takes the > 64 KB dynamic-LDS instantiation"""
    from eprecon_amd import dense2d as D2
    from eprecon_amd.modules import Conv2d_Block, Conv2d_Residual_Block
    torch.manual_seed(cin * 100 + cout)
    dev = _dev()
    v, h, w = 9, 50, 70
    a = Conv2d_Block(cin, cmid, 3).to(dev).train()
    b = Conv2d_Block(cmid, cout, 3).to(dev).train()
    r = Conv2d_Residual_Block(cout, 3).to(dev).train()
    for m in (a, b, r):
        for prm in m.parameters():
            if prm.dim() == 1:
                prm.data.uniform_(0.5, 1.5)
    x = torch.randn(v, cin, h, w, device=dev)
    g = D2.PixelGrid.get(v, h, w, dev)
    with torch.no_grad():
        ref = r(b(a(x)))
        act = a.run_act(D2.Act(D2.rows_of(_cl(x))), g)
        act = b.run_act(act, g)
        act = r.run_act(act, g)
        got = D2.maps_of(D2.materialize(act), v, h, w)
    assert (got - ref).abs().max().item() < TOL


@pytest.mark.parametrize("cin,cmid,cout", [(12, 12, 12), (24, 12, 20), (24, 24, 24), (40, 40, 40), (20, 36, 8)])
def test_direct_gather_kernel_chain_on_the_pixel_map(monkeypatch, cin, cmid, cout):
    """the same chain with the 3x3 layers on the direct gather kernel (spconv_direct16_kernel over the pixel map: what the
    9 x 120 x 160 level of the fusion stack runs on): zero padding at the image borders = missing neighbours, which must stay
    zero AFTER the producer's pending BatchNorm is applied on load; C_in = 12 / 24 / 40 end in a half chunk"""
    from eprecon_amd import dense2d as D2
    from eprecon_amd.modules import Conv2d_Block, Conv2d_Residual_Block
    from test_sparse_gpu import _last_conv_kernel
    monkeypatch.setattr(D2, "DIRECT_2D_MIN_ROWS", 1000)
    torch.manual_seed(cin * 100 + cout + 7)
    dev = _dev()
    v, h, w = 9, 50, 70
    a = Conv2d_Block(cin, cmid, 3).to(dev).train()
    b = Conv2d_Block(cmid, cout, 3).to(dev).train()
    r = Conv2d_Residual_Block(cout, 3).to(dev).train()
    for m in (a, b, r):
        for prm in m.parameters():
            if prm.dim() == 1:
                prm.data.uniform_(0.5, 1.5)
    x = torch.randn(v, cin, h, w, device=dev)
    g = D2.PixelGrid.get(v, h, w, dev)
    with torch.no_grad():
        ref = r(b(a(x)))
        act, name = _last_conv_kernel((9, cin, cmid, 1000), lambda: a.run_act(D2.Act(D2.rows_of(_cl(x))), g))
        assert name == "spconv_direct16_kernel"
        act = b.run_act(act, g)
        act = r.run_act(act, g)
        got = D2.maps_of(D2.materialize(act), v, h, w)
    assert (got - ref).abs().max().item() < TOL
