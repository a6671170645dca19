"""GPU: init -> coarse selection and upsample through the C ABI, bit-exact against the oracle and
the reference golden vectors."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import grid_ops as OG  # noqa: E402
from test_oracle_grid_ops import selection_inputs  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "grid_ops.npz"))


def test_init_select_matches_reference(gold):
    from eprecon_amd.grid_ops import init_select
    logit, coords = selection_inputs(gold)
    got, per_batch = init_select(dev(logit), dev(coords), 1)
    assert np.array_equal(got.cpu().numpy()[:, 1:], gold["sel_coords"])
    assert per_batch == [len(gold["sel_coords"])]


def test_init_select_two_batches_and_empty():
    from eprecon_amd.grid_ops import init_select
    rng = np.random.default_rng(4)
    xyz = np.argwhere(np.ones((48, 48, 48), bool)) * 2
    coords = np.concatenate([np.concatenate([np.full((len(xyz), 1), b), xyz], 1) for b in (0, 1)]).astype(np.int32)
    gx = coords[:, 1] / 10.0
    logit = (np.sin(gx) * 2 + rng.standard_normal(len(coords)) * 0.3).astype(np.float32)
    logit[coords[:, 0] == 1] -= 0.7
    got, per_batch = init_select(dev(logit), dev(coords), 2)
    ref = OG.init_select(logit, coords, 2)
    assert np.array_equal(got.cpu().numpy(), ref)
    assert per_batch == [int((ref[:, 0] == 0).sum()), int((ref[:, 0] == 1).sum())]
    got, per_batch = init_select(dev(np.full(len(coords), -9.0, np.float32)), dev(coords), 2)
    assert got.shape[0] == 0 and per_batch == [0, 0]


def test_upsample(gold):
    from eprecon_amd.grid_ops import upsample
    rng = np.random.default_rng(31)
    coords = rng.integers(0, 24, size=(500, 3)) * 4
    coords = np.concatenate([rng.integers(0, 2, size=(500, 1)), coords], 1).astype(np.int32)
    feat = rng.standard_normal((500, 7)).astype(np.float32)
    uf, uc = upsample(dev(feat), dev(coords), 2)
    assert np.array_equal(uf.cpu().numpy(), gold["up_feat"])
    assert np.array_equal(uc.cpu().numpy(), gold["up_coords"])
    big_c = rng.integers(0, 96, size=(60000, 4)).astype(np.int32)
    big_f = rng.standard_normal((60000, 50)).astype(np.float32)
    uf, uc = upsample(dev(big_f), dev(big_c), 1)
    rf, rc = OG.upsample(big_f, big_c, 1)
    assert np.array_equal(uf.cpu().numpy(), rf) and np.array_equal(uc.cpu().numpy(), rc)
