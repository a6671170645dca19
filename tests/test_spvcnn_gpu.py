"""GPU parity of the point-voxel path: aligned-camera coordinates, voxelise / devoxelise, SPVCNN and
the sparse ConvGRU, against the numpy oracle (restated torchsparse semantics: parity unpinned).
Integer results (voxel coords, point->voxel ids, corner indices) bit-exact; features within 1e-3."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from eprecon_amd import synthetic as S  # noqa: E402
from oracle import pointvoxel as PV  # noqa: E402
from oracle import sparse as OS  # noqa: E402
from oracle import spvcnn as ON  # noqa: E402

TOL = 1e-3


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def shell_coords(seed, interval, n_target, batch=1):
    """voxels near the analytic surface of the synthetic scene, raster order, (b,x,y,z) finest units"""
    window = S.make_window(seed=seed)
    lvl = {4: 2, 2: 1, 1: 0}[interval]
    tsdf = S.analytic_tsdf(window, lvl)
    xyz = np.argwhere(np.abs(tsdf) < 0.9) * interval
    rng = np.random.default_rng(seed)
    if len(xyz) > n_target:
        xyz = xyz[np.sort(rng.choice(len(xyz), n_target, replace=False))]
    rows = [np.concatenate([np.full((len(xyz), 1), b), xyz], 1) for b in range(batch)]
    return window, np.concatenate(rows).astype(np.int32)


def test_aligned_coords_bit_exact():
    from eprecon_amd.torchsparse_utils import aligned_camera_coords
    window, coords = shell_coords(1, 2, 30000, batch=2)
    origin = np.stack([window["vol_origin_partial"], window["vol_origin_partial"] + 0.08]).astype(np.float32)
    w2ac = np.stack([window["world_to_aligned_camera"]] * 2)
    got = aligned_camera_coords(dev(coords), dev(origin), 0.04, dev(w2ac)).cpu().numpy()
    assert np.array_equal(got, PV.aligned_coords(coords, origin, 0.04, w2ac))


@pytest.mark.parametrize("interval,vres", [(4, 0.16), (1, 0.04)])
def test_voxelize_devoxelize(interval, vres):
    from eprecon_amd.tensor import PointTensor
    from eprecon_amd.torchsparse_utils import initial_voxelize, point_to_voxel, voxel_to_point
    window, coords = shell_coords(2, interval, 20000)
    pts = PV.aligned_coords(coords, window["vol_origin_partial"][None], 0.04, window["world_to_aligned_camera"][None])
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((len(pts), 12)).astype(np.float32)
    z = PointTensor(dev(feat), dev(pts))
    x = initial_voxelize(z, 1, vres)
    zo = PV.Points(feat, pts)
    c0, f0, inv = PV.initial_voxelize(zo, 1, vres)
    assert np.array_equal(x.C.cpu().numpy(), c0)
    assert np.array_equal(z.additional_features["idx_query"][1].cpu().numpy(), inv)
    assert np.array_equal(z.C.cpu().numpy(), zo.C) and np.array_equal(z.vox.cpu().numpy(), zo.vox)
    assert np.abs(x.F.cpu().numpy() - f0).max() < 1e-5
    assert len(c0) < len(pts)  # the rotated frame merges some voxels
    for stride, vset in ((1, x.vset), (2, x.vset.downsample()[0])):
        vc = vset.coords.cpu().numpy()
        idx, w = PV.trilinear(vc, stride, zo.C)
        from eprecon_amd.tensor import SparseTensor
        vf = rng.standard_normal((len(vc), 12)).astype(np.float32)
        zp = voxel_to_point(SparseTensor(dev(vf), vset), z)
        assert np.array_equal(z.idx_query[stride].cpu().numpy(), idx)
        assert np.abs(z.weights[stride].cpu().numpy() - w).max() < 1e-6
        assert np.abs(zp.F.cpu().numpy() - PV.devoxelize(vf, idx, w)).max() < 1e-4
        back = point_to_voxel(SparseTensor(dev(vf), vset), z)
        assert np.abs(back.F.cpu().numpy() - PV.point_to_voxel(vc, stride, zo, feat)).max() < 1e-5


def _sd(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


@pytest.mark.parametrize("stage,cin,n", [(0, 80, 6000), (1, 138, 12000), (2, 74, 20000)])
def test_spvcnn_matches_oracle(stage, cin, n):
    from eprecon_amd.modules import SPVCNN
    from eprecon_amd.tensor import PointTensor
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
        out = net(PointTensor(dev(feat), dev(pts))).cpu().numpy()
        # the pass issued by ONE library call (default) and launch by launch from Python: the same launches, bit for bit
        import eprecon_amd.modules as M
        assert M._NATIVE_SPVCNN and net._native[1] is not None
        M._NATIVE_SPVCNN = False
        try:
            py = net(PointTensor(dev(feat), dev(pts))).cpu().numpy()
        finally:
            M._NATIVE_SPVCNN = True
        assert np.array_equal(out, py)
    ref = ON.spvcnn_forward(_sd(net), feat, pts, 1, vres)
    assert out.shape == ref.shape == (len(pts), int(96 * cr))
    err = np.abs(out - ref).max()
    assert err < TOL, err


def test_native_pass_follows_weight_rewrites():
    """the one-call SPVCNN pass keeps packed copies of every weight.  (a) a write through `p.data` (dist.broadcast,
    EMA swap) + sparse.clear_packed_weights and (b) a REPLACED Parameter object (load_state_dict(assign=True)) must both reach
    the native pass: it stays bit-identical to the Python-issued pass, and differs from its own result before the write."""
    import eprecon_amd.modules as M
    from eprecon_amd import sparse as SP
    from eprecon_amd.modules import SPVCNN
    from eprecon_amd.tensor import PointTensor
    window, coords = shell_coords(3, 4, 6000)
    pts = PV.aligned_coords(coords, window["vol_origin_partial"][None], 0.04, window["world_to_aligned_camera"][None])
    feat = np.random.default_rng(5).standard_normal((len(pts), 80)).astype(np.float32)
    torch.manual_seed(5)
    net = SPVCNN(num_classes=1, in_channels=80, pres=1, cr=1.0, vres=0.16, dropout=False).cuda()

    def both():
        with torch.no_grad():
            nat = net(PointTensor(dev(feat), dev(pts))).cpu().numpy()
            assert net._native[1] is not None
            M._NATIVE_SPVCNN = False
            try:
                py = net(PointTensor(dev(feat), dev(pts))).cpu().numpy()
            finally:
                M._NATIVE_SPVCNN = True
        assert np.array_equal(nat, py)
        return nat

    first = both()
    with torch.no_grad():      # (a) in-place through .data: neither the version counter nor the pointer moves
        for p_ in net.parameters():
            if p_.dim() >= 2:
                p_.data.mul_(1.25)
    SP.clear_packed_weights(net)
    second = both()
    assert not np.array_equal(first, second)
    # (b) new Parameter objects (same values scaled again), no clear call: the key holds id() of the live objects
    sd = {k: (v * 0.5 if v.dim() >= 2 else v).clone() for k, v in net.state_dict().items()}
    net.load_state_dict(sd, assign=True)
    third = both()
    assert not np.array_equal(second, third)


def test_sphash_order_matches_oracle():
    """torchsparse's voxel order (ascending F.sphash): HIP hash == numpy restatement, negative coordinates
    included; the sorted order is what ConvGRU's stale-index reuse depends on"""
    from eprecon_amd import _lib
    rng = np.random.default_rng(12)
    c = np.unique(np.concatenate([np.zeros((5000, 1), np.int64), rng.integers(-700, 2500, (5000, 3))], 1), axis=0)
    c = c.astype(np.int32)
    lib = _lib.load()
    ct = dev(c)
    h = torch.empty(len(c), dtype=torch.int64, device="cuda")
    _lib.check(lib.eprecon_sphash_async(_lib.ptr(ct), len(c), _lib.ptr(h), _lib.current_stream()), "sphash")
    ref = PV.sphash(c)
    assert np.array_equal(h.cpu().numpy(), ref) and (ref >= 0).all() and len(np.unique(ref)) == len(ref)
    # the whole order in one call (hash + bucket sort + inverse permutation, csrc/hash_order.hip) == argsort of the oracle's hashes
    perm = torch.empty(len(c), dtype=torch.int32, device="cuda")
    rank = torch.empty(len(c), dtype=torch.int32, device="cuda")
    ws = torch.empty(lib.eprecon_sphash_order_workspace_bytes(len(c)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.eprecon_sphash_order_async(_lib.ptr(ct), len(c), _lib.ptr(perm), _lib.ptr(rank), _lib.ptr(ws), ws.numel(),
                                              _lib.current_stream()), "sphash_order")
    order = np.argsort(ref, kind="stable")
    assert np.array_equal(perm.cpu().numpy(), order)
    assert np.array_equal(rank.cpu().numpy()[order], np.arange(len(c)))


@pytest.mark.parametrize("n", [1, 7, 130, 4097, 70000, 400000])
def test_sphash_order_at_every_bucket_count(n):
    """the bucket sort of csrc/hash_order.hip from one voxel to more voxels than its 32768 buckets x 8; rows repeated on purpose
    (equal hashes: the order between them is the row order, as a stable sort gives)"""
    from eprecon_amd import _lib
    rng = np.random.default_rng(n)
    c = np.concatenate([np.zeros((n, 1), np.int64), rng.integers(-3000, 3000, (n, 3))], 1).astype(np.int32)
    if n > 100:
        c[n // 2:n // 2 + 50] = c[:50]          # duplicates -> ties
    lib = _lib.load()
    ct = dev(c)
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    rank = torch.empty(n, dtype=torch.int32, device="cuda")
    ws = torch.empty(lib.eprecon_sphash_order_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    for _ in range(2):      # (twice through the same workspace: the call clears what it counts in)
        _lib.check(lib.eprecon_sphash_order_async(_lib.ptr(ct), n, _lib.ptr(perm), _lib.ptr(rank), _lib.ptr(ws), ws.numel(),
                                                  _lib.current_stream()), "sphash_order")
    order = np.argsort(PV.sphash(c), kind="stable")
    assert np.array_equal(perm.cpu().numpy(), order)
    assert np.array_equal(rank.cpu().numpy()[order], np.arange(n))


@pytest.mark.parametrize("literal", [True, False])
@pytest.mark.parametrize("ch,scale", [(96, 0), (24, 2)])
def test_convgru_matches_oracle(ch, scale, literal, monkeypatch):
    """literal: convr devoxelises with convz's cached corner indices into its own (finer) voxel set in
    torchsparse's hash order, as the reference does (ops/torchsparse_utils.py:70-71,97-99)"""
    from eprecon_amd import torchsparse_utils as TU
    from eprecon_amd.modules import ConvGRU
    from eprecon_amd.tensor import PointTensor
    monkeypatch.setattr(TU, "LITERAL_CONVR", literal)
    TU.clear_voxelization_cache()
    interval, vres = 2 ** (2 - scale), 0.04 * 2 ** (2 - scale)
    window, coords = shell_coords(7 + scale, interval, 8000)
    pts = PV.aligned_coords(coords, window["vol_origin_partial"][None], 0.04, window["world_to_aligned_camera"][None])
    pts[:, 3] = 0
    rng = np.random.default_rng(ch)
    h = rng.standard_normal((len(pts), ch)).astype(np.float32)
    x = rng.standard_normal((len(pts), ch)).astype(np.float32)
    torch.manual_seed(ch)
    gru = ConvGRU(hidden_dim=ch, input_dim=ch, pres=1, vres=vres).cuda()
    with torch.no_grad():
        coords_t = dev(pts)
        out = gru(PointTensor(dev(h), coords_t), PointTensor(dev(x), coords_t)).cpu().numpy()
    sd = {"g." + k: v for k, v in _sd(gru).items()}
    ref = ON.convgru(sd, "g", h, x, pts, 1, vres, literal=literal)
    err = np.abs(out - ref).max()
    assert err < TOL, err
    other = ON.convgru(sd, "g", h, x, pts, 1, vres, literal=not literal)
    assert np.abs(other - ref).max() > 10 * TOL      # the two behaviours are distinguishable
