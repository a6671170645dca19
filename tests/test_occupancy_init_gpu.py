"""GPU: Occupancy_Initialization.forward (HIP variance volume + submanifold stack) against the
oracle pipeline on the same seeded weights; selection of the stage-0 voxels bit-exact given the
same logits."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from eprecon_amd import synthetic as S  # noqa: E402
from oracle import back_project as OB  # noqa: E402
from oracle import grid_ops as OG  # noqa: E402
from oracle import occupancy_init as OI  # noqa: E402


def make_inputs(seed, height, width, n_vox, batch=1):
    window = S.make_window(seed=seed, width=width, height=height, n_vox=n_vox)
    shapes = S.pyramid_shapes(height, width)
    rng = np.random.default_rng(seed + 100)
    feats = []
    for v in range(9):
        feats.append([torch.from_numpy(rng.standard_normal((batch,) + shapes[l], dtype=np.float32)).cuda()
                      for l in range(3)])
    return window, feats


# (480, 640, 96, 1) is BASELINE.json configs[1]: 9 x 640x480 views, dense 48^3 grid = 110,592 voxels
@pytest.mark.parametrize("height,width,nvox,batch", [(240, 320, 48, 1), (240, 320, 48, 2), (480, 640, 96, 1)])
def test_occupancy_initialization_matches_oracle(height, width, nvox, batch):
    from eprecon_amd.occupancy_initialization import Occupancy_Initialization
    torch.manual_seed(0)
    net = Occupancy_Initialization([80, 40, 24], 32, 9).cuda()
    with torch.no_grad():  # non-trivial affine parameters
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    n_vox = (nvox,) * 3
    window, feats = make_inputs(3, height, width, n_vox, batch)
    coords = S.dense_coords(n_vox, 2, batch=batch)
    origin = np.repeat(window["vol_origin_partial"][None], batch, 0).copy()
    if batch > 1:
        origin[1, 0] += 0.2
    kr = np.ascontiguousarray(np.repeat(window["proj_matrices"][:, 1][:, None], batch, 1))
    shape = tuple(n // 2 for n in n_vox)
    with torch.no_grad():
        out = net(torch.from_numpy(coords).cuda(), torch.from_numpy(origin).cuda(), 0.04, feats,
                  torch.from_numpy(kr).cuda(), shape, 1, 2)
        assert out is not None
        occ, coord_init, count = out
        fused = torch.stack([net.feat_fusion_pre(torch.stack([f[2][b] for f in feats]),
                                                 torch.stack([f[1][b] for f in feats]),
                                                 torch.stack([f[0][b] for f in feats]))
                             for b in range(batch)], dim=1).cpu().numpy()
    ref = OB.back_project(coords, origin, 0.04, fused, kr, 2, OB.MODE_VARIANCE)
    assert np.array_equal(coord_init.cpu().numpy(), ref["coords"])
    assert np.array_equal(count.cpu().numpy(), ref["count"])
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    logits = []
    for b in range(batch):
        m = ref["coords"][:, 0] == b
        logits.append(OI.sparse_stack(sd, ref["feats"][m], ref["coords"][m], 2))
    ref_logit = np.concatenate(logits)
    got = occ.cpu().numpy()
    assert got.shape == ref_logit.shape
    # the logits are batch-normalised (unit variance): 1e-3 absolute == north_star tolerance
    assert np.abs(got - ref_logit).max() < 1e-3, np.abs(got - ref_logit).max()
    # stage-0 selection from the HIP logits == oracle selection from the same logits
    from eprecon_amd.grid_ops import init_select
    sel, per_batch = init_select(occ, coord_init, batch, dim=shape[0] // 2, cell=4)
    assert np.array_equal(sel.cpu().numpy(), OG.init_select(got, ref["coords"], batch, dim=shape[0] // 2))


def test_returns_none_below_1000_valid_voxels():
    from eprecon_amd.occupancy_initialization import Occupancy_Initialization
    net = Occupancy_Initialization([80, 40, 24], 32, 9).cuda()
    window, feats = make_inputs(5, 240, 320, (16, 16, 16))
    coords = S.dense_coords((16, 16, 16), 2)  # 512 voxels < 1000
    kr = np.ascontiguousarray(window["proj_matrices"][:, 1][:, None])
    with torch.no_grad():
        out = net(torch.from_numpy(coords).cuda(), torch.from_numpy(window["vol_origin_partial"][None]).cuda(),
                  0.04, feats, torch.from_numpy(kr).cuda(), (8, 8, 8), 1, 2)
    assert out is None


def test_hip_dense_path_matches_pytorch_path():
    """feat_fusion_pre on the GPU inference path (channels-last, HIP BatchNorm2d / upsampling, HIP
    graph replay) against the plain PyTorch modules (grad-enabled path)"""
    from eprecon_amd.occupancy_initialization import Occupancy_Initialization
    torch.manual_seed(3)
    net = Occupancy_Initialization([80, 40, 24], 32, 9).cuda()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    rng = np.random.default_rng(0)
    f1 = torch.from_numpy(rng.standard_normal((9, 80, 15, 20), dtype=np.float32)).cuda()
    f2 = torch.from_numpy(rng.standard_normal((9, 40, 30, 40), dtype=np.float32)).cuda()
    f4 = torch.from_numpy(rng.standard_normal((9, 24, 60, 80), dtype=np.float32)).cuda()
    ref = net.feat_fusion_pre(f1, f2, f4).detach()          # grad enabled -> PyTorch BN / interpolate
    with torch.no_grad():
        got = net.feat_fusion_pre(f1, f2, f4)
        assert got.is_contiguous(memory_format=torch.channels_last)
        views = [list(t.unbind(0)) for t in (f1, f2, f4)]
        graphed = net._fusion_graphed(views).clone()
        graphed2 = net._fusion_graphed(views)
    assert (got - ref).abs().max().item() < 2e-4
    assert (graphed - ref).abs().max().item() < 2e-4 and torch.equal(graphed, graphed2)
    # round 6: the graph's inputs are the pixel rows, written from the 27 per-view maps by ONE launch (eprecon_views_to_rows_async)
    # instead of three torch.stack launches + three channels-last copies: the same rows, the same stack -> the same bits
    assert torch.equal(graphed, got)
    with torch.no_grad():   # maps that are not contiguous float32 take the stacked-NCHW inputs as before
        odd = [[m.transpose(1, 2).contiguous().transpose(1, 2) for m in v] for v in views]
        assert not odd[0][0].is_contiguous()
        assert torch.equal(net._fusion_graphed(odd), got)
