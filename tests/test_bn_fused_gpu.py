"""BatchNorm statistics finished INSIDE the producing convolution (bn_finalize_in_kernel, csrc/sparse_conv.hip): summaries
stored write-through, two levels of arrival counters, the last arrivers merge in row / group order.  Checked against the
separate finalize launch on the summaries of the same launch (another merge order -> 1e-5), for every kernel of the family,
with the workspace reused across many launches of changing size (stale counters or stale L1 lines would show here)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


class Owner:   # stands in for the layer object that keeps the workspace
    pass


@pytest.fixture(autouse=True)
def fused_on(monkeypatch):
    """the in-kernel finalize is off by default (EPRECON_BN_TICKET, eprecon_amd/sparse.py): these tests turn it on"""
    from eprecon_amd import sparse as SP
    monkeypatch.setattr(SP, "FUSED_FINALIZE", True)


def reference_affine(partial, gamma, beta, eps):
    from eprecon_amd import sparse as SP
    return SP.bn_affine(partial, gamma, beta, eps)


def check(out, partial, aff, gamma, beta, eps, n):
    assert aff is not None
    torch.cuda.synchronize()
    assert abs(float(partial[:, 0, 0].sum()) - n) < 0.5
    s_ref, t_ref = reference_affine(partial, gamma, beta, eps)
    assert torch.allclose(aff[0], s_ref, rtol=2e-5, atol=1e-6), float((aff[0] - s_ref).abs().max())
    assert torch.allclose(aff[1], t_ref, rtol=2e-5, atol=2e-5), float((aff[1] - t_ref).abs().max())
    # and against the definition on the stored tensor
    y = out.double()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    sc = gamma.double() / torch.sqrt(var + eps)
    assert torch.allclose(aff[0].double(), sc, rtol=1e-4, atol=1e-6)
    assert torch.allclose(aff[1].double(), beta.double() - mean * sc, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,k,cin,cout", [
    (100, 27, 16, 16),         # one summary row
    (5000, 27, 32, 32),        # resident kernel, 40 rows (3 groups)
    (60000, 27, 32, 64),       # two column tiles per wave
    (94000, 27, 48, 24),
    (3000, 27, 128, 128),      # split-K (32-row summary blocks), 128 columns = 4 column blocks
    (20000, 1, 96, 128),       # per-point linear layer (identity map)
    (70000, 8, 64, 64),        # strided map shape
    (33000, 27, 80, 40),       # wide input
    (203000, 27, 16, 8),       # SPVCNN's finest level: > 1,008 summary rows, narrow output
])
def test_finalize_inside_the_gather_kernels(n, k, cin, cout):
    from eprecon_amd import sparse as SP
    g = torch.Generator(device="cuda").manual_seed(n + cout)
    gamma = torch.rand(cout, device="cuda", generator=g) + 0.5
    beta = torch.randn(cout, device="cuda", generator=g)
    owner = Owner()
    for rep in range(6):     # the workspace is reused: counters must be back at zero, sizes change
        m = n if rep % 2 == 1 else max(1, n // 3 + 17 * rep)   # short launches first: their group rows must not reach the counters
        x = torch.randn((m, cin), device="cuda", generator=g) * (1 + rep)
        w = torch.randn((k, cin, cout), device="cuda", generator=g) / (k * cin) ** 0.5
        nbr = None
        if k > 1:
            nbr = torch.randint(-1, m, (k, m), device="cuda", generator=g, dtype=torch.int32)
        out, partial, aff = SP.conv_stats(x, w, nbr, bn=(gamma, beta, 1e-5), owner=owner)
        check(out, partial, aff, gamma, beta, 1e-5, m)


def test_finalize_inside_the_image_tile_kernel_and_hip_graph_replay():
    """the 2D fusion stack's path (dense2d.conv_bn_launch): tile kernel for narrow 3x3 layers, gather form otherwise, both
    replayed from a captured HIP graph (the counters must come back to zero after every replay)"""
    import torch.nn as nn
    from eprecon_amd import dense2d as D2
    D2._FUSED_FINALIZE = True      # (off by default for the 2D stack: EPRECON_BN_TICKET_2D)
    torch.manual_seed(0)
    for cin, cout, ks, (v, h, w) in ((24, 24, 3, (9, 120, 160)), (40, 20, 3, (9, 60, 80)), (80, 80, 1, (9, 30, 40)),
                                      (96, 24, 1, (9, 120, 160))):
        conv, bn = nn.Conv2d(cin, cout, ks, padding="same").cuda(), nn.BatchNorm2d(cout).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
        grid = D2.PixelGrid.get(v, h, w, torch.device("cuda"))
        x = torch.randn((v * h * w, cin), device="cuda")
        with torch.no_grad():
            for _ in range(2):
                a = D2.conv_bn_act(conv, bn, D2.Act(x), grid, relu=True)
            got = D2.materialize(a)
            ref = torch.relu(bn(conv(D2.maps_of(x, v, h, w).contiguous())))        # train-mode BatchNorm2d
            assert float((D2.maps_of(got, v, h, w) - ref).abs().max()) < 1e-3
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                D2.conv_bn_act(conv, bn, D2.Act(x), grid, relu=True)
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(graph):
                a = D2.conv_bn_act(conv, bn, D2.Act(x), grid, relu=True)
                rows = D2.materialize(a)
            for rep in range(4):
                x.copy_(torch.randn_like(x) * (rep + 1))
                graph.replay()
                ref = torch.relu(bn(conv(D2.maps_of(x, v, h, w).contiguous())))
                assert float((D2.maps_of(rows, v, h, w) - ref).abs().max()) < 1e-3 * (rep + 1)
    D2._FUSED_FINALIZE = False


def test_finalize_inside_the_dense_grid_kernels(monkeypatch):
    from eprecon_amd import sparse as SP
    from test_dense_conv3d_gpu import dev, grid_set
    monkeypatch.setenv("EPRECON_CONV_DENSE3D", "3")
    rng = np.random.default_rng(5)
    for dims, fill, cin, cout in (((48, 48, 48), 0.85, 32, 1), ((20, 14, 24), 0.6, 16, 16), ((48, 48, 48), 0.85, 32, 32)):
        c = grid_set(rng, dims, 2, fill)
        vs = SP.VoxelSet(dev(c), 2, dims=dims)
        dm = SP.DenseMap(vs, dims)
        owner = Owner()
        gamma, beta = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        for rep in range(3):
            x = torch.randn((len(c), cin), device="cuda") * (rep + 1)
            w = torch.randn((27, cin, cout), device="cuda") / (27 * cin) ** 0.5
            out, partial, aff = SP.conv_stats(x, w, dm, bn=(gamma, beta, 1e-5), owner=owner)
            check(out, partial, aff, gamma, beta, 1e-5, len(c))


def test_spvcnn_block_outputs_do_not_depend_on_where_the_batchnorm_is_finished(monkeypatch):
    """a residual block with the statistics finished in the launches == the same block with separate finalize launches"""
    from eprecon_amd import modules as M
    from eprecon_amd import sparse as SP
    from test_oracle_sparse import random_coords
    rng = np.random.default_rng(2)
    c = random_coords(rng, 30000, extent=40, batch=1)
    vs = SP.VoxelSet(torch.from_numpy(c).cuda(), 1)
    torch.manual_seed(3)
    blk = M.ResidualBlock(32, 64).cuda()
    x = torch.randn((vs.n, 32), device="cuda")
    with torch.no_grad():
        a = blk.run(x, vs.kernel_map(3)).clone()
        monkeypatch.setattr(SP, "FUSED_FINALIZE", False)
        b = blk.run(x, vs.kernel_map(3))
    assert float((a - b).abs().max()) < 1e-4
