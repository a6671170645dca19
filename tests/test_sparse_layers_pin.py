"""The reference pin of the torchsparse / spconv layers, READY TO FIRE ("parity unpinned").

tests/golden/make_golden.py gen_sparse_layers() dumps per-layer activations of the reference's own SPVCNN
(models/modules.py:75-175, ops/torchsparse_utils.py:15-105), ConvGRU / SConv3d (:178-222) and SparseSubMConv3d (:249-271) with
seeded weights — wherever `torchsparse` and `spconv` import.  They do not in the build container (no wheel, no source, no
network), so tests/golden/sparse_layers.npz does not exist and the GPU tests below SKIP with that reason; the CPU tests check
that the generator reaches its "skipped" exit here and that the seeded inputs are machine-independent.  The day the fixture
exists, these tests compare this package's HIP layers with it: outputs to 1e-3 (north_star), the voxel set of every voxel-side
layer as a SET (torchsparse numbers voxels by ascending hash: order unspecified), per-layer activations matched by coordinate.
This is also the check of checkpoint parity (offset enumeration order, weight layout) README.md calls unverified."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURE = os.path.join(GOLDEN, "sparse_layers.npz")
ABSENT = "tests/golden/sparse_layers.npz absent: torchsparse / spconv were not importable where the fixtures were generated " \
         "(parity of the sparse layers is unpinned; python tests/golden/make_golden.py sparse_layers writes it where they are)"
TOL = 1e-3


def _inputs():
    sys.path.insert(0, GOLDEN)
    from cases import sparse_layer_inputs
    return sparse_layer_inputs()


def test_inputs_are_seeded_and_machine_independent():
    a, b = _inputs(), _inputs()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["pts"].shape == (6000, 4) and a["pts"].dtype == np.float32 and np.all(a["pts"][:, 3] == 0)
    assert a["sub_coords"].dtype == np.int32 and len(np.unique(a["sub_coords"], axis=0)) == len(a["sub_coords"])
    # a checksum of the generator's stream: a numpy whose default_rng produced other inputs would silently un-pin the fixture
    assert abs(float(a["feats"].astype(np.float64).sum()) - float(b["feats"].astype(np.float64).sum())) == 0.0


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="build container only (the generator imports /root/reference)")
def test_generator_runs_to_its_skipped_exit_or_writes_the_fixture():
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "sparse_layers"], capture_output=True, text=True,
                       timeout=600, cwd=GOLDEN)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "sparse_layers: skipped" in r.stdout or os.path.exists(FIXTURE), r.stdout


def _by_coords(coords):
    """row order of a coordinate list, for order-insensitive comparison"""
    c = np.asarray(coords).astype(np.int64)
    return np.lexsort(tuple(c[:, k] for k in range(c.shape[1] - 1, -1, -1)))


@pytest.fixture(scope="module")
def gold():
    if not os.path.exists(FIXTURE):
        pytest.skip(ABSENT)
    return np.load(FIXTURE)


def _load(module, gold, prefix):
    import torch
    sd = {k[len(prefix):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected and all("running_" in m or "num_batches" in m for m in missing), (missing, unexpected)


@pytest.mark.gpu
def test_spvcnn_matches_the_reference_layers(gold):
    import torch
    from eprecon_amd.modules import SPVCNN
    from eprecon_amd.tensor import PointTensor
    inp = _inputs()
    dev = torch.device("cuda")
    net = SPVCNN(num_classes=1, in_channels=inp["feats"].shape[1], pres=1, cr=0.25, vres=0.04, dropout=False).train()
    _load(net, gold, "spvcnn/sd/")
    net = net.to(dev)
    with torch.no_grad():
        y = net(PointTensor(torch.from_numpy(inp["feats"]).to(dev), torch.from_numpy(inp["pts"]).to(dev)))
    assert np.abs(y.cpu().numpy() - gold["spvcnn/out"]).max() < TOL     # per-point output: point order is the input's
    # voxel sets the reference visited, as sets: stem (stride 1), the two down stages (strides 2, 4); x-y-z-batch columns
    from eprecon_amd import torchsparse_utils as TU
    z = PointTensor(torch.from_numpy(inp["feats"]).to(dev), torch.from_numpy(inp["pts"]).to(dev))
    with torch.no_grad():
        x0 = TU.initial_voxelize(z, 1, 0.04, levels=3)
    s1 = x0.vset
    s2, _, _ = s1.downsample()
    s4, _, _ = s2.downsample()
    for key, vs in (("spvcnn/stem.0/C", s1), ("spvcnn/stage1.0.net.0/C", s2), ("spvcnn/stage2.0.net.0/C", s4)):
        if key in gold.files:
            ours = vs.coords.cpu().numpy()[:, [1, 2, 3, 0]]
            ref = gold[key][:, :4]
            assert {tuple(r) for r in ours.tolist()} == {tuple(r) for r in ref.astype(np.int64).tolist()}, key


@pytest.mark.gpu
def test_convgru_matches_the_reference_cell(gold):
    import torch
    from eprecon_amd.modules import ConvGRU
    from eprecon_amd.tensor import PointTensor
    inp = _inputs()
    dev = torch.device("cuda")
    gru = ConvGRU(hidden_dim=12, input_dim=12, pres=1, vres=0.04).train()
    _load(gru, gold, "convgru/sd/")
    gru = gru.to(dev)
    pts = torch.from_numpy(inp["pts"]).to(dev)
    with torch.no_grad():
        h = gru(PointTensor(torch.from_numpy(inp["h"]).to(dev), pts), PointTensor(torch.from_numpy(inp["x"]).to(dev), pts.clone()))
    assert np.abs(h.cpu().numpy() - gold["convgru/out"]).max() < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("k", [3, 1])
def test_submanifold_conv_matches_spconv(gold, k):
    import torch
    from eprecon_amd.modules import SparseSubMConv3d
    inp = _inputs()
    dev = torch.device("cuda")
    conv = SparseSubMConv3d(16, 8, k)
    conv.load_spconv_weight(torch.from_numpy(gold[f"subm{k}/weight"]))
    with torch.no_grad():
        conv.bias.copy_(torch.from_numpy(gold[f"subm{k}/bias"]))
    conv = conv.to(dev)
    with torch.no_grad():
        y = conv(torch.from_numpy(inp["sub_feats"]).to(dev), torch.from_numpy(inp["sub_coords"]).to(dev), [24, 24, 24], 1)
    assert np.abs(y.cpu().numpy() - gold[f"subm{k}/out"]).max() < TOL
