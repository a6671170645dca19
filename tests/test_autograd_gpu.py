"""f4: gradients of the HIP operators (eprecon_amd/autograd.py) against plain PyTorch references of the same ops.
fp32 both sides; the accumulation orders differ, tolerance 2e-4 relative to the gradient's scale."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import torch.nn.functional as F  # noqa: E402
from test_oracle_sparse import random_coords  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, tol=2e-4):
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err < tol, err


def ref_conv(x, w, nbr, bias):
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    if nbr is None:
        y = x @ w3[0]
    else:
        pad = torch.cat([x, x.new_zeros(1, x.shape[1])])
        idx = torch.where(nbr >= 0, nbr, torch.full_like(nbr, x.shape[0])).long()
        y = torch.einsum("kni,kio->no", pad[idx], w3)
    return y if bias is None else y + bias


def conv_case(n, cin, cout, kind, seed):
    from eprecon_amd.sparse import VoxelSet
    rng = np.random.default_rng(seed)
    c = random_coords(rng, n, extent=18, batch=2)
    vs = VoxelSet(dev(c), 1)
    if kind == "k3":
        nbr, n_in, n_out, kvol = vs.kernel_map(3), vs.n, vs.n, 27
    elif kind == "down":
        coarse, down, up = vs.downsample()
        nbr, n_in, n_out, kvol = down, vs.n, coarse.n, 8
    elif kind == "up":
        coarse, down, up = vs.downsample()
        nbr, n_in, n_out, kvol = up, coarse.n, vs.n, 8
    else:
        nbr, n_in, n_out, kvol = None, vs.n, vs.n, 1
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(n_in, cin, device="cuda", generator=g)
    w = torch.randn((kvol, cin, cout) if kvol > 1 else (cin, cout), device="cuda", generator=g) / np.sqrt(cin * kvol)
    b = torch.randn(cout, device="cuda", generator=g)
    dy = torch.randn(n_out, cout, device="cuda", generator=g)
    return x, w, b, nbr, dy


@pytest.mark.parametrize("n,cin,cout,kind", [(3000, 32, 32, "k3"), (5000, 81, 32, "k3"), (2500, 139, 64, "k3"), (4000, 128, 128, "k3"),
                                             (1500, 320, 128, "k3"), (6000, 32, 32, "down"), (6000, 64, 64, "down"),
                                             (6000, 128, 96, "up"), (6000, 96, 96, "up"), (7000, 32, 128, "lin"), (333, 176, 1, "lin"),
                                             (1, 8, 8, "k3"), (3000, 74, 8, "k3"), (3000, 8, 74, "k3"), (3000, 51, 12, "k3"),
                                             (3000, 16, 16, "k3"), (3000, 24, 40, "k3")])
def test_sparse_conv_gradients(n, cin, cout, kind):
    from eprecon_amd import autograd as AG
    x, w, b, nbr, dy = conv_case(n, cin, cout, kind, seed=n + cin)
    got, ref = [], []
    for fn, sink in ((AG.sparse_conv, got), (lambda x_, w_, nbr_, b_: ref_conv(x_, w_, nbr_, b_), ref)):
        xs, ws, bs = (t.clone().requires_grad_() for t in (x, w, b))
        y = fn(xs, ws, nbr, bs)
        y.backward(dy)
        sink.extend([y.detach(), xs.grad, ws.grad, bs.grad])
    for a, r in zip(got, ref):
        close(a, r)


def test_weight_gradient_is_deterministic():
    from eprecon_amd import autograd as AG
    x, w, b, nbr, dy = conv_case(20000, 64, 64, "k3", seed=9)
    a = AG.conv_weight_grad(x, dy, nbr, 27, 64, 64)
    c = AG.conv_weight_grad(x, dy, nbr, 27, 64, 64)
    assert torch.equal(a, c)


def test_inverse_map_is_the_transpose():
    from eprecon_amd import autograd as AG
    _, _, _, nbr, _ = conv_case(4000, 8, 8, "k3", seed=2)
    inv = AG.inverse_map(nbr, nbr.shape[1])
    assert torch.equal(inv, nbr.flip(0))            # a submanifold map is symmetric: offset k <-> offset 26 - k
    _, _, _, down, _ = conv_case(4000, 8, 8, "down", seed=2)
    _, _, _, up, _ = conv_case(4000, 8, 8, "up", seed=2)
    assert torch.equal(AG.inverse_map(down, up.shape[1]), up)


def point_case(n, c, seed):
    from eprecon_amd.tensor import PointTensor
    from eprecon_amd.torchsparse_utils import initial_voxelize, voxel_to_point, clear_voxelization_cache
    clear_voxelization_cache()
    g = torch.Generator(device="cuda").manual_seed(seed)
    pts = torch.rand(n, 4, device="cuda", generator=g) * 12
    pts[:, 3] = 0
    z = PointTensor(torch.randn(n, c, device="cuda", generator=g), pts)
    with torch.no_grad():
        x = initial_voxelize(z, 1, 1.7)
        voxel_to_point(x, z)          # fills z.idx_query / z.weights
    return z, x


def test_devoxelize_and_segment_mean_gradients():
    from eprecon_amd import autograd as AG
    z, x = point_case(6000, 37, seed=4)
    idx8, w8 = z.idx_query[1], z.weights[1]
    m = x.F.shape[0]
    g = torch.Generator(device="cuda").manual_seed(1)
    vf = torch.randn(m, 37, device="cuda", generator=g)
    dout = torch.randn(6000, 37, device="cuda", generator=g)
    a = vf.clone().requires_grad_()
    AG.devoxelize(a, idx8, w8).backward(dout)
    r = vf.clone().requires_grad_()
    pad = torch.cat([r, r.new_zeros(1, 37)])
    idx = torch.where(idx8 >= 0, idx8, torch.full_like(idx8, m)).long()
    ((pad[idx] * w8[:, :, None]).sum(1)).backward(dout)
    close(a.grad, r.grad)

    inverse, lists = z.additional_features["idx_query"][1], z.additional_features["lists"][1]
    dvox = torch.randn(m, 37, device="cuda", generator=g)
    a = z.F.clone().requires_grad_()
    out = AG.segment_mean(a, inverse, lists, m)
    out.backward(dvox)
    r = z.F.clone().requires_grad_()
    counts = torch.bincount(inverse.long(), minlength=m).float()
    ref = torch.zeros(m, 37, device="cuda").index_add_(0, inverse.long(), r) / counts[:, None]
    ref.backward(dvox)
    close(out.detach(), ref.detach())
    close(a.grad, r.grad)


def test_devoxelize_gradient_is_bit_identical_run_to_run():
    """no float atomics: the (point, corner) entries of a voxel are summed in CSR-list order"""
    from eprecon_amd import autograd as AG
    z, x = point_case(50000, 24, seed=7)
    idx8, w8 = z.idx_query[1], z.weights[1]
    m = x.F.shape[0]
    g = torch.Generator(device="cuda").manual_seed(3)
    vf = torch.randn(m, 24, device="cuda", generator=g)
    dout = torch.randn(50000, 24, device="cuda", generator=g)
    grads = []
    for _ in range(3):
        a = vf.clone().requires_grad_()
        AG.devoxelize(a, idx8, w8).backward(dout)
        grads.append(a.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


def _grid_sample_reference(feats, grid, mask, mode):
    """feats [V,B=1,C,H,W], grid [V,N,2], mask bool[V,N] -> the reference's masked mean / variance"""
    v = feats.shape[0]
    samples = F.grid_sample(feats[:, 0], grid.view(v, 1, -1, 2), padding_mode="zeros", align_corners=True)[:, :, 0]   # [V,C,N]
    samples = samples * mask[:, None, :]
    cnt = mask.sum(0).clamp(min=1).float()
    mean = samples.sum(0) / cnt
    if mode == "mean":
        return mean.t(), None
    var = (((samples - mean[None]) * mask[:, None, :]) ** 2).sum(0) / cnt
    return var.t(), mean.t()


@pytest.mark.parametrize("mode,c", [("mean", 24), ("mean_depth", 40), ("variance", 32)])
def test_back_project_gradients(mode, c):
    from eprecon_amd import autograd as AG
    from eprecon_amd import back_project as BP
    from eprecon_amd import synthetic as S
    window = S.make_window(seed=4)
    coords = dev(S.dense_coords((96, 96, 96), 4))
    feats = dev(S.make_features(9, 9, (c, 30, 40)))
    krcam = dev(window["proj_matrices"][:, 2][:, None])
    origin, vs = dev(window["vol_origin_partial"][None]), 0.04
    hip_mode = {"mean": BP.MODE_MEAN, "mean_depth": BP.MODE_MEAN_DEPTH, "variance": BP.MODE_VARIANCE}[mode]
    with torch.no_grad():
        aux = BP.run(coords, origin, vs, feats, krcam, 2, hip_mode, want_grid=True)
    a = feats.clone().requires_grad_()
    res = AG.back_project(coords, origin, vs, a, krcam, 2, hip_mode, want_mean=(mode == "variance"))
    g = torch.Generator(device="cuda").manual_seed(3)
    dout = torch.randn(res["feats"].shape, device="cuda", generator=g)
    dmean = torch.randn(res["n_valid"], c, device="cuda", generator=g)
    loss = (res["feats"] * dout).sum() + ((res["mean"] * dmean).sum() if mode == "variance" else 0)
    loss.backward()
    r = feats.clone().requires_grad_()
    out, mean = _grid_sample_reference(r, aux["grid"], aux["mask"], "variance" if mode == "variance" else "mean")
    loss = (out * dout[:, :c]).sum() + ((mean * dmean).sum() if mode == "variance" else 0)
    loss.backward()
    close(res["feats"][:, :c].detach(), out.detach(), 1e-4)
    close(a.grad, r.grad, 5e-4)


def _spvcnn_case(stage=1, cin=138, n=6000):
    from eprecon_amd import synthetic as S  # noqa: F401
    from eprecon_amd.modules import SPVCNN
    from oracle import pointvoxel as PV
    from test_spvcnn_gpu import shell_coords
    interval, vres, cr = 2 ** (2 - stage), 0.04 * 2 ** (2 - stage), 1 / 2 ** stage
    window, coords = shell_coords(3 + stage, interval, n)
    pts = PV.aligned_coords(coords, window["vol_origin_partial"][None], 0.04, window["world_to_aligned_camera"][None])
    rng = np.random.default_rng(stage)
    feat = rng.standard_normal((len(pts), cin)).astype(np.float32)
    torch.manual_seed(stage)
    net = SPVCNN(num_classes=1, in_channels=cin, pres=1, cr=cr, vres=vres, dropout=False).cuda()
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    return net, feat, pts, vres


def test_spvcnn_recording_forward_matches_oracle_and_inference():
    from eprecon_amd.tensor import PointTensor
    from oracle import spvcnn as ON
    net, feat, pts, vres = _spvcnn_case()
    out = net(PointTensor(dev(feat), dev(pts)))
    assert out.requires_grad
    ref = ON.spvcnn_forward({k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}, feat, pts, 1, vres)
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 1e-3
    with torch.no_grad():
        inf = net(PointTensor(dev(feat), dev(pts)))
    close(out.detach(), inf, 1e-4)


def _torch_devoxelize(feat, idx8, w8):
    pad = torch.cat([feat, feat.new_zeros(1, feat.shape[1])])
    idx = torch.where(idx8 >= 0, idx8, torch.full_like(idx8, feat.shape[0])).long()
    return (pad[idx] * w8[:, :, None]).sum(1)


def _torch_segment_mean(feat, idx, lists, m):
    keep = idx >= 0
    counts = torch.bincount(idx[keep].long(), minlength=m).float().clamp(min=1)
    return torch.zeros(m, feat.shape[1], device=feat.device).index_add_(0, idx[keep].long(), feat[keep]) / counts[:, None]


def _grads_of(run, params, inputs):
    for t in params + inputs:
        t.grad = None
    run().backward()
    return [t.grad.clone() for t in params + inputs]


def test_spvcnn_gradients_match_a_pure_torch_composition(monkeypatch):
    """the whole point-voxel U-Net twice through the SAME module code: once on the HIP Functions, once with every
    Function swapped for a dense PyTorch expression of the same operator; all parameter and input gradients agree"""
    from eprecon_amd import autograd as AG
    from eprecon_amd.tensor import PointTensor
    net, feat, pts, vres = _spvcnn_case(stage=2, cin=74, n=3000)
    x = dev(feat).requires_grad_()
    g = torch.Generator(device="cuda").manual_seed(5)
    probe = torch.randn(len(pts), 24, device="cuda", generator=g)

    def run():
        return (torch.tanh(net(PointTensor(x, dev(pts)))) * probe).sum()

    params = list(net.parameters())
    hip = _grads_of(run, params, [x])
    monkeypatch.setattr(AG, "sparse_conv", lambda x_, w_, nbr_=None, bias_=None: ref_conv(x_, w_, nbr_, bias_))
    monkeypatch.setattr(AG, "devoxelize", _torch_devoxelize)
    monkeypatch.setattr(AG, "segment_mean", _torch_segment_mean)
    ref = _grads_of(run, params, [x])
    names = [n for n, _ in net.named_parameters()] + ["input"]
    for name, a, r in zip(names, hip, ref):
        scale = max(float(r.abs().max()), 5e-2)      # (a Linear bias in front of a BatchNorm has an exactly-zero gradient)
        assert float((a - r).abs().max()) / scale < 2e-3, name


def test_convgru_recording_matches_oracle_and_trains():
    from eprecon_amd import torchsparse_utils as TU
    from eprecon_amd.modules import ConvGRU
    from eprecon_amd.tensor import PointTensor
    from oracle import pointvoxel as PV
    from oracle import spvcnn as ON
    from test_spvcnn_gpu import shell_coords
    TU.clear_voxelization_cache()
    ch, scale = 24, 2
    interval, vres = 2 ** (2 - scale), 0.04 * 2 ** (2 - scale)
    window, coords = shell_coords(7 + scale, interval, 5000)
    pts = PV.aligned_coords(coords, window["vol_origin_partial"][None], 0.04, window["world_to_aligned_camera"][None])
    pts[:, 3] = 0
    rng = np.random.default_rng(ch)
    h = rng.standard_normal((len(pts), ch)).astype(np.float32)
    x = rng.standard_normal((len(pts), ch)).astype(np.float32)
    torch.manual_seed(ch)
    gru = ConvGRU(hidden_dim=ch, input_dim=ch, pres=1, vres=vres).cuda()
    coords_t = dev(pts)
    ht, xt = dev(h).requires_grad_(), dev(x).requires_grad_()
    out = gru(PointTensor(ht, coords_t), PointTensor(xt, coords_t))
    sd = {"g." + k: v.detach().cpu().numpy() for k, v in gru.state_dict().items()}
    ref = ON.convgru(sd, "g", h, x, pts, 1, vres, literal=TU.LITERAL_CONVR)
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 1e-3
    out.square().mean().backward()
    for t in [ht, xt] + list(gru.parameters()):
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().sum()) > 0


@pytest.mark.parametrize("n,cin,cout", [(94000, 32, 32), (140000, 48, 24)])
def test_conv_adjoint_identities_at_full_size(n, cin, cout):
    """size-independent property at BASELINE's list sizes: the backward operators are the adjoints of the forward one,
    <conv(x; W), dy> = <x, dgrad(dy; W)> = <W, wgrad(x, dy)>  (fp32 sums of ~10^7 products: 2e-4 relative)"""
    from eprecon_amd import autograd as AG
    from eprecon_amd import synthetic as S
    from eprecon_amd.sparse import VoxelSet
    rng = np.random.default_rng(n)
    coords = S.dense_coords((96, 96, 96), 2 if n < 110592 else 1)
    keep = np.sort(rng.choice(len(coords), n, replace=False))
    vs = VoxelSet(dev(coords[keep]), 2 if n < 110592 else 1)
    nbr = vs.kernel_map(3)
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(n, cin, device="cuda", generator=g).requires_grad_()
    w = (torch.randn(27, cin, cout, device="cuda", generator=g) / np.sqrt(27 * cin)).requires_grad_()
    dy = torch.randn(n, cout, device="cuda", generator=g)
    y = AG.sparse_conv(x, w, nbr)
    y.backward(dy)
    lhs = float((y.detach().double() * dy.double()).sum())
    via_x = float((x.detach().double() * x.grad.double()).sum())
    via_w = float((w.detach().double() * w.grad.double()).sum())
    scale = float((y.detach().double() * dy.double()).abs().sum())
    assert abs(lhs - via_x) < 2e-4 * scale and abs(lhs - via_w) < 2e-4 * scale, (lhs, via_x, via_w, scale)


@pytest.mark.parametrize("kind,cin,cout", [("k3", 48, 24), ("down", 32, 32), ("up", 64, 48)])
def test_conv_backward_matches_numpy_oracle(kind, cin, cout):
    """HIP dgrad / wgrad / inverted map against oracle/backward.py (the adjoint written out in numpy)"""
    from eprecon_amd import autograd as AG
    from oracle import backward as OB
    x, w, b, nbr, dy = conv_case(3000, cin, cout, kind, seed=17)
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    AG.sparse_conv(xs, ws, nbr).backward(dy)
    nbr_h = nbr.cpu().numpy()
    assert np.array_equal(AG.inverse_map(nbr, x.shape[0]).cpu().numpy(), OB.invert_map(nbr_h, x.shape[0]))
    dx, dw, _ = OB.conv_backward(x.cpu().numpy(), w.cpu().numpy(), nbr_h, dy.cpu().numpy())
    close(xs.grad.cpu(), torch.from_numpy(dx))
    close(ws.grad.cpu(), torch.from_numpy(dw))
