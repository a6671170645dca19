"""Host-side layout logic of the round-4 entry points (no GPU): the operand order the one-launch heads kernel expects from
sparse._pack_mlp_weight (include/eprecon_hip.h: eprecon_mlp4x_async), the tap-major depthwise weights, and the sizing
functions of the library that do not touch a device."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("k,m", [(24, 96), (96, 24), (88, 352), (48, 1), (176, 48)])
def test_mlp_weight_packing_matches_the_header(k, m):
    """block (t, c), lane 16 q + j, component i = Wt[16 c + 4 q + i][16 t + j], zero outside the matrix"""
    from eprecon_amd.sparse import _pack_mlp_weight
    torch.manual_seed(k + m)
    wt = torch.randn(k, m)
    packed = _pack_mlp_weight(wt).numpy()
    kc, mt = (k + 15) // 16, (m + 15) // 16
    assert packed.shape == (mt * kc * 64 * 4,)
    p = packed.reshape(mt, kc, 4, 16, 4)            # [t][c][q][j][i]
    w = np.zeros((16 * kc, 16 * mt), np.float32)
    w[:k, :m] = wt.numpy()
    rng = np.random.default_rng(0)
    for _ in range(200):
        t, c, q, j, i = rng.integers(mt), rng.integers(kc), rng.integers(4), rng.integers(16), rng.integers(4)
        assert p[t, c, q, j, i] == w[16 * c + 4 * q + i, 16 * t + j]
    assert np.count_nonzero(packed) == np.count_nonzero(wt.numpy())


def test_pack_mlp4x_views_are_aligned_and_cached():
    from eprecon_amd.modules import Linear4xTrans
    from eprecon_amd.sparse import clear_packed_weights, pack_mlp4x
    m = Linear4xTrans(24, 1)
    a = pack_mlp4x(m)
    assert set(a) == {"w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "w3", "b3"}
    assert all(v.data_ptr() % 16 == 0 and v.numel() % 16 == 0 for v in a.values())
    assert a["b3"].numel() == 16 and float(a["b3"][1:].abs().sum()) == 0.0          # C_out = 1 padded to a tile
    assert a["g2"].numel() == 32 and float(a["g2"][24:].abs().sum()) == 0.0         # C = 24 padded to two tiles, zeros behind
    assert pack_mlp4x(m) is a                        # cached per parameter version
    with torch.no_grad():
        m.linear3.weight.add_(1.0)                   # an in-place write bumps the version
    assert pack_mlp4x(m) is not a
    b = pack_mlp4x(m)
    m.linear3.weight.data.mul_(2.0)                  # a write through .data does not: clear_packed_weights is the contract
    assert pack_mlp4x(m) is b
    clear_packed_weights(m)
    assert pack_mlp4x(m) is not b


def test_depthwise_taps_are_tap_major():
    import torch.nn as nn
    from eprecon_amd.backbone import _dw_taps, _is_depthwise
    conv = nn.Conv2d(8, 8, 5, padding=2, stride=2, groups=8, bias=False)
    taps = _dw_taps(conv)
    assert taps.shape == (25, 8) and taps.is_contiguous()
    assert torch.equal(taps[7], conv.weight.detach()[:, 0, 1, 2])        # tap (dy, dx) = (1, 2) -> row 5 * 1 + 2
    assert _is_depthwise(conv)
    assert not _is_depthwise(nn.Conv2d(8, 8, 3, padding=1))              # dense
    assert not _is_depthwise(nn.Conv2d(8, 8, 3, padding=0, groups=8, bias=False))    # not 'same' padded
    assert not _is_depthwise(nn.Conv2d(6, 6, 3, padding=1, groups=6, bias=False))    # channels not a multiple of 4


def test_sizing_functions_of_the_library():
    from eprecon_amd import _lib
    lib = _lib.load()
    # per-view BatchNorm: ~16 rows per row lane, between 1 and 128 ranges per view
    assert lib.eprecon_bn2d_views_chunks(76800, 32) == 128 and lib.eprecon_bn2d_views_chunks(1200, 480) == 38
    assert lib.eprecon_bn2d_views_chunks(10, 16) == 1
    assert lib.eprecon_bn2d_views_workspace_bytes(9, 1200, 480) == 9 * 38 * 3 * 480 * 4
    # shapes the one-launch heads take
    assert lib.eprecon_mlp4x_supported(96, 1) and lib.eprecon_mlp4x_supported(176, 48) and lib.eprecon_mlp4x_supported(48, 48)
    assert not lib.eprecon_mlp4x_supported(96, 48) and not lib.eprecon_mlp4x_supported(50, 1)
    # workspaces grow with their arguments and are 0 for nothing to do
    assert lib.eprecon_sphash_order_workspace_bytes(0) == 0
    assert lib.eprecon_sphash_order_workspace_bytes(400000) > lib.eprecon_sphash_order_workspace_bytes(1000) > 0
    assert lib.eprecon_gru_stage_finish_workspace_bytes(1000, 900, 1000) >= lib.eprecon_sphash_order_workspace_bytes(1000)
    assert lib.eprecon_spvcnn_geometry_workspace_bytes(1000, 900, 100) > 0


def test_backbone_walker_on_the_pytorch_fallbacks():
    """MnasMulti._run_hip (one module of look-ahead: pending BatchNorm in front of a depthwise layer, the block's skip added by
    the last BatchNorm's apply pass) walks the trunk correctly: on CPU tensors every HIP piece falls back to its PyTorch
    expression, and the result must equal the plain per-view route"""
    import eprecon_amd.backbone as BB
    torch.manual_seed(0)
    net = BB.MnasMulti(1.0).train()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.7, 1.3)
                m.bias.uniform_(-0.2, 0.2)
        x = torch.randn(4, 3, 32, 48).contiguous(memory_format=torch.channels_last)     # two views of two images
        v = 2
        for stage in (net.conv0, net.conv1):
            a = net._run(stage, x, v)
            b = net._run_hip(stage, x, v)
            assert a.shape == b.shape
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-4), float((a - b).abs().max())
            x = a
