"""GPU: the HIP kernels against vectors captured from the reference's own lines (round-2 pins):
view mean / variance volume (G3), aligned-camera coordinates (a7), the 2D fusion stack as a whole
(G6), and the dense-grid generator (a1) — all through the C ABI / the product modules."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cases  # noqa: E402
from test_oracle_occ_init import check_variance, fusion_pre_module  # noqa: E402


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "occ_init.npz"))


@pytest.fixture(scope="module")
def gold_ac(golden_dir):
    return np.load(os.path.join(golden_dir, "aligned_coords.npz"))


@pytest.mark.parametrize("name", list(cases.OCC_INIT_CASES))
@pytest.mark.parametrize("channels_last", [False, True])
def test_hip_variance_matches_reference(gold, name, channels_last):
    from eprecon_amd import back_project as BP
    window, coords, origin, fused, kr = cases.occ_init_case(name)
    f = _dev(fused)
    if channels_last:   # the layout the fused maps arrive in on the product path
        f = f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
    res = BP.view_variance(_dev(coords), _dev(origin), window["voxel_size"], f, _dev(kr), 2)
    assert res is not None
    check_variance(gold, name, res["count"].cpu().numpy(), res["coords"].cpu().numpy(),
                   res["var"].cpu().numpy(), res["mean"].cpu().numpy(), 1e-3)


@pytest.mark.parametrize("scale", [0, 1, 2])
def test_hip_aligned_coords_match_reference(gold_ac, scale):
    from eprecon_amd.torchsparse_utils import aligned_camera_coords
    coords, origin, w2ac, interval = cases.aligned_case(scale)
    got = aligned_camera_coords(_dev(coords), _dev(origin), 0.04, _dev(w2ac)).cpu().numpy()
    assert np.array_equal(got, gold_ac[f"s{scale}_r_coords"])        # models/neucon_network.py:387-398
    for i in range(2):                                               # models/gru_fusion.py:332-337
        c = coords[coords[:, 0] == i].copy()
        c[:, 0] = 0
        g = aligned_camera_coords(_dev(c), _dev(origin[i:i + 1]), 0.04, _dev(w2ac[i:i + 1])).cpu().numpy()
        assert np.array_equal(g, gold_ac[f"s{scale}_b{i}_gru_r_coords"])


def test_hip_feat_fusion_pre_matches_reference(gold):
    """the HIP rows path (gather-GEMM convolutions, pending BatchNorms, HIP graph) against the
    reference's own feat_fusion_pre output on identical seeded weights"""
    net = fusion_pre_module(gold).cuda()
    f1, f2, f4 = (_dev(a) for a in cases.fusion_pre_inputs())
    ref = gold["fusion_pre_out"]
    with torch.no_grad():
        y = net.feat_fusion_pre(f1, f2, f4)
        assert y.is_contiguous(memory_format=torch.channels_last)     # i.e. the rows path ran
        g = net._fusion_graphed([list(t.unbind(0)) for t in (f1, f2, f4)]).clone()
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-3
    assert np.abs(g.cpu().numpy() - ref).max() < 1e-3


def test_generate_grid_on_device(golden_dir):
    """eprecon_amd.generate_grids on cuda:0 against ops/generate_grids.py:3-10 (grid_ops.npz) and the
    raster order every dense-grid test feeds (synthetic.dense_coords)"""
    from eprecon_amd import synthetic as S
    from eprecon_amd.generate_grids import dense_coords, generate_grid
    gold = np.load(os.path.join(golden_dir, "grid_ops.npz"))
    dev = torch.device("cuda")
    for interval in (1, 2, 4):
        g, dims = generate_grid([96, 96, 96], interval, device=dev)
        assert g.is_cuda and g.dtype == torch.float32
        a = g.cpu().numpy().astype(np.float64)
        assert tuple(dims) == tuple(gold[f"grid_i{interval}_dims"])
        assert np.array_equal(g.cpu().numpy()[:, :200], gold[f"grid_i{interval}_head"])
        chk = np.array([a.sum(), (a * np.arange(1, a.shape[1] + 1)).sum()])
        assert np.array_equal(chk, gold[f"grid_i{interval}_checksum"])
        for batch in (1, 2):
            c, d2 = dense_coords([96, 96, 96], interval, batch, device=dev)
            assert c.is_cuda and c.dtype == torch.int32 and tuple(d2) == tuple(dims)
            assert np.array_equal(c.cpu().numpy(), S.dense_coords((96, 96, 96), interval, batch=batch))
