"""The per-voxel heads as one launch (csrc/heads.hip, eprecon_mlp4x_async) against the PyTorch modules of
Linear4xTrans (models/modules.py:273-311) on the same rows: fp32 round-off (a different summation order), every
channel count NeuConNet uses, ragged row counts and row pitches, the shared launch of the TSDF / occupancy pair."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(mod, x):
    import torch.nn.functional as F
    h = F.relu(mod.norm1(mod.linear1(x)))
    h = F.relu(mod.norm2(mod.linear2(h)))
    y = mod.linear3(h)
    return y + h if mod.use_residual else y


def _module(cin, cout, seed):
    from eprecon_amd.modules import Linear4xTrans
    torch.manual_seed(seed)
    m = Linear4xTrans(cin, cout).cuda()
    with torch.no_grad():       # non-trivial biases / LayerNorm affines (the default init has zeros and ones)
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.3)
    return m


@pytest.mark.parametrize("cin,cout", [(24, 1), (48, 1), (96, 1), (48, 48), (88, 48), (176, 48)])
@pytest.mark.parametrize("n,extra", [(1, 0), (63, 3), (1000, 81), (20011, 1), (50001, 0)])   # (<= 40,000 rows: four waves per 16 voxels)
def test_one_launch_equals_the_modules(cin, cout, n, extra):
    from eprecon_amd import sparse as SP
    m = _module(cin, cout, 1)
    torch.manual_seed(n)
    wide = torch.randn((n, cin + extra), device="cuda") * 2.0
    x = wide[:, :cin]                       # a column slice: row pitch cin + extra (16-byte aligned rows or not)
    with torch.no_grad():
        assert SP.mlp4x_supported(cin, cout)
        y = m(x)
        ref = _reference(m, x.contiguous())
    assert y.shape == (n, cout)
    assert torch.allclose(y, ref, rtol=2e-4, atol=2e-4), float((y - ref).abs().max())


def test_tsdf_and_occupancy_share_a_launch():
    from eprecon_amd import _lib
    from eprecon_amd.modules import linear4x_pair
    a, b = _module(96, 1, 2), _module(96, 1, 3)
    x = torch.randn((5000, 96), device="cuda")
    with torch.no_grad():
        ya, yb = linear4x_pair(a, b, x)
        assert torch.allclose(ya, _reference(a, x), rtol=2e-4, atol=2e-4)
        assert torch.allclose(yb, _reference(b, x), rtol=2e-4, atol=2e-4)
        # deterministic: a second launch gives the same bits
        ya2, yb2 = linear4x_pair(a, b, x)
    assert torch.equal(ya, ya2) and torch.equal(yb, yb2)


def test_weight_writes_are_seen_after_clear(monkeypatch):
    """the packed copy is cached per (version, data_ptr); writes through .data need clear_packed_weights (like every other
    operand-order cache of the package)"""
    from eprecon_amd import sparse as SP
    m = _module(24, 1, 4)
    x = torch.randn((100, 24), device="cuda")
    with torch.no_grad():
        y0 = m(x)
        m.linear3.weight.data.mul_(2.0)
        SP.clear_packed_weights(m)
        y1 = m(x)
        ref = _reference(m, x)
    assert torch.allclose(y1, ref, rtol=2e-4, atol=2e-4) and not torch.allclose(y0, y1)


def test_switch_off_runs_the_modules(monkeypatch):
    import eprecon_amd.modules as M
    m = _module(48, 1, 5)
    x = torch.randn((300, 48), device="cuda")
    with torch.no_grad():
        y_on = m(x)
        monkeypatch.setattr(M, "_FUSED_HEADS", False)
        y_off = m(x)
    assert torch.allclose(y_on, y_off, rtol=2e-4, atol=2e-4)
