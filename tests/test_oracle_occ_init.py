"""CPU: pins for the occupancy initialiser and the aligned-camera coordinates against vectors
captured from the reference's own lines (tests/golden/make_golden.py: occ_init, aligned_coords):
  * oracle/c MODE_VARIANCE (view mean / population variance, valid set, counts) vs
    Occupancy_Initialization.forward run up to its first spconv line      
  * eprecon_oracle_aligned_coords vs models/neucon_network.py:387-398 and models/gru_fusion.py:332-337
  * feat_fusion_pre (the PyTorch modules of this package on CPU = what cpu_baseline times) vs the
    reference's own feat_fusion_pre with identical seeded weights          """
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cases  # noqa: E402

from oracle import back_project as O  # noqa: E402
from oracle import pointvoxel as PV  # noqa: E402


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "occ_init.npz"))


@pytest.fixture(scope="module")
def gold_ac(golden_dir):
    return np.load(os.path.join(golden_dir, "aligned_coords.npz"))


def check_variance(gold, name, count, coords, var, mean, tol):
    """shared with the GPU test: counts / valid set bit exact, statistics within `tol`"""
    assert np.array_equal(np.asarray(count).astype(np.uint8), gold[name + "_count"])
    assert var.shape[0] == int(gold[name + "_nvalid"])
    rows = gold[name + "_rows"]
    sub = coords[rows].copy()
    sub[:, 1:] //= 2
    sub[:, 0] = 0
    assert np.array_equal(sub, gold[name + "_subm_coord_rows"])   # :131-133 of the reference
    assert np.abs(var[rows] - gold[name + "_var_rows"]).max() < tol
    assert np.abs(mean[rows] - gold[name + "_mean_rows"]).max() < tol
    rs = lambda a: a.sum(1, dtype=np.float64).astype(np.float32)[::cases.ROW_STRIDE]
    np.testing.assert_allclose(rs(var), gold[name + "_var_rowsum"], atol=32 * tol, rtol=0)
    np.testing.assert_allclose(rs(mean), gold[name + "_mean_rowsum"], atol=32 * tol, rtol=0)


@pytest.mark.parametrize("name", list(cases.OCC_INIT_CASES))
def test_oracle_variance_matches_reference(gold, name):
    window, coords, origin, fused, kr = cases.occ_init_case(name)
    res = O.back_project(coords, origin, window["voxel_size"], fused, kr, 2, O.MODE_VARIANCE)
    check_variance(gold, name, res["count"], res["coords"], res["feats"], res["mean"], 1e-5)


@pytest.mark.parametrize("scale", [0, 1, 2])
def test_oracle_aligned_coords_match_reference(gold_ac, scale):
    coords, origin, w2ac, interval = cases.aligned_case(scale)
    got = PV.aligned_coords(coords, origin, 0.04, w2ac)
    ref = gold_ac[f"s{scale}_r_coords"]
    assert got.shape == ref.shape and np.array_equal(got[:, 3], ref[:, 3])
    assert np.array_equal(got, ref)     # bit exact: the k-ordered fma chain == torch's CPU matmul
    for i in range(2):
        m = coords[:, 0] == i
        c = coords[m].copy()
        c[:, 0] = 0
        g = PV.aligned_coords(c, origin[i:i + 1], 0.04, w2ac[i:i + 1])
        r = gold_ac[f"s{scale}_b{i}_gru_r_coords"]
        assert np.array_equal(g, r) and not g[:, 3].any()


def fusion_pre_module(gold):
    import torch
    from eprecon_amd.occupancy_initialization import Occupancy_Initialization
    torch.manual_seed(0)
    net = Occupancy_Initialization(list(cases.FUSION_PRE_CH), cases.FUSION_PRE_DOWN, 9)
    cases.seeded_state(net, 2024, keys=[str(k) for k in gold["fusion_pre_keys"]])
    net.train()
    return net


def test_feat_fusion_pre_matches_reference(gold):
    import torch
    net = fusion_pre_module(gold)
    f1, f2, f4 = (torch.from_numpy(a) for a in cases.fusion_pre_inputs())
    with torch.no_grad():
        y = net.feat_fusion_pre(f1, f2, f4).numpy()
    ref = gold["fusion_pre_out"]
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < 1e-4, np.abs(y - ref).max()


def test_torch_cpu_sparse_stack_equals_the_numpy_oracle():
    """oracle/torch_cpu.py (what bench.py's cpu_baseline times for the spconv layers: gather -> matmul -> index_add_ per
    offset on the torch CPU threads) against the numpy restatement of the same wiring (oracle/occupancy_init.py)"""
    import torch
    from eprecon_amd.config import CH_IMG, CH_INIT_DOWN, N_VIEWS
    from eprecon_amd.occupancy_initialization import Occupancy_Initialization
    from oracle import occupancy_init as OI
    from oracle import torch_cpu as TC
    torch.manual_seed(5)
    net = Occupancy_Initialization(CH_IMG, CH_INIT_DOWN, N_VIEWS).train()
    for prm in net.parameters():
        if prm.dim() == 1:
            prm.data.normal_(0.5, 0.3)
    sd_t = {k: v.detach() for k, v in net.state_dict().items()}
    sd_n = {k: v.numpy() for k, v in sd_t.items()}
    rng = np.random.default_rng(3)
    g = np.stack(np.meshgrid(np.arange(14), np.arange(12), np.arange(10), indexing="ij"), -1).reshape(-1, 3) * 2
    g = g[rng.random(len(g)) < 0.8]
    coords = np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)
    var = rng.standard_normal((len(coords), 32)).astype(np.float32)
    ref = OI.sparse_stack(sd_n, var, coords, 2)
    with torch.no_grad():
        got = TC.sparse_stack(sd_t, torch.from_numpy(var), TC.kernel_map_pairs(coords, 2)).numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-4
