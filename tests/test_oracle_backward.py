"""f4: the numpy restatement of the sparse operators' backward (oracle/backward.py) against torch autograd of dense
formulations of the forward operators, on CPU."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import backward as OB  # noqa: E402
from oracle import sparse as OS  # noqa: E402
from test_oracle_sparse import random_coords  # noqa: E402


def _maps(rng):
    c = random_coords(rng, 900, extent=12, batch=2)
    k3 = OS.kernel_map(c, c, 3, 1)
    coarse, parent = OS.unique_first(c, 2)
    down = OS.kernel_map(c, coarse, 2, 1)
    up = OS.transpose_map(c, parent, 1)
    return {"k3": (k3, len(c), len(c)), "down": (down, len(c), len(coarse)), "up": (up, len(coarse), len(c))}


@pytest.mark.parametrize("kind,cin,cout", [("k3", 7, 5), ("down", 4, 6), ("up", 6, 3)])
def test_conv_backward_matches_autograd(kind, cin, cout):
    rng = np.random.default_rng(3)
    nbr, n_in, n_out = _maps(rng)[kind]
    x = rng.standard_normal((n_in, cin)).astype(np.float32)
    w = rng.standard_normal((nbr.shape[0], cin, cout)).astype(np.float32)
    dy = rng.standard_normal((n_out, cout)).astype(np.float32)
    xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
    bt = torch.zeros(cout, requires_grad=True)
    idx = torch.from_numpy(np.where(nbr >= 0, nbr, n_in).astype(np.int64))
    pad = torch.cat([xt, torch.zeros(1, cin)])
    y = torch.einsum("kni,kio->no", pad[idx], wt) + bt
    assert np.abs(y.detach().numpy() - OS.sparse_conv(x, nbr, w)).max() < 1e-4
    y.backward(torch.from_numpy(dy))
    dx, dw, db = OB.conv_backward(x, w, nbr, dy, bias=True)
    for got, ref in ((dx, xt.grad), (dw, wt.grad), (db, bt.grad)):
        assert np.abs(got - ref.numpy()).max() < 1e-3 * max(float(ref.abs().max()), 1.0)


def test_inverse_of_a_submanifold_map_is_its_mirror():
    rng = np.random.default_rng(5)
    nbr, n, _ = _maps(rng)["k3"]
    assert np.array_equal(OB.invert_map(nbr, n), nbr[::-1])
    maps = _maps(np.random.default_rng(5))
    assert np.array_equal(OB.invert_map(maps["down"][0], maps["down"][1]), maps["up"][0])


def test_point_voxel_transfers_backward():
    rng = np.random.default_rng(8)
    m, n, c = 300, 1000, 6
    idx8 = rng.integers(-1, m, (n, 8)).astype(np.int32)
    w8 = rng.random((n, 8)).astype(np.float32)
    dout = rng.standard_normal((n, c)).astype(np.float32)
    feat = torch.randn(m, c, requires_grad=True)
    pad = torch.cat([feat, torch.zeros(1, c)])
    out = (pad[torch.from_numpy(np.where(idx8 >= 0, idx8, m).astype(np.int64))] * torch.from_numpy(w8)[:, :, None]).sum(1)
    out.backward(torch.from_numpy(dout))
    assert np.abs(OB.devoxelize_backward(dout, idx8, w8, m) - feat.grad.numpy()).max() < 1e-4

    idx = rng.integers(-1, m, n).astype(np.int32)
    pf = torch.randn(n, c, requires_grad=True)
    keep = torch.from_numpy(idx >= 0)
    counts = torch.bincount(torch.from_numpy(idx[idx >= 0].astype(np.int64)), minlength=m).float().clamp(min=1)
    mean = torch.zeros(m, c).index_add_(0, torch.from_numpy(idx[idx >= 0].astype(np.int64)), pf[keep]) / counts[:, None]
    dmean = rng.standard_normal((m, c)).astype(np.float32)
    mean.backward(torch.from_numpy(dmean))
    assert np.abs(OB.segment_mean_backward(dmean, idx, m) - pf.grad.numpy()).max() < 1e-5
