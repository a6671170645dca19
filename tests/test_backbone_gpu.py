"""The 2D feeder's HIP layers (csrc/backbone2d.hip) against PyTorch: depthwise convolutions against F.conv2d in float64 on
the CPU, the per-view train-mode BatchNorm against F.instance_norm / per-view nn.BatchNorm2d, and MnasMulti.forward_views
against V separate forward() calls (the reference's call pattern, models/neuralrecon.py:53-54)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,s", [(3, 1), (3, 2), (5, 1), (5, 2)])
@pytest.mark.parametrize("c,h,w", [(32, 30, 40), (72, 17, 23), (480, 6, 9)])
def test_depthwise_conv_equals_conv2d(k, s, c, h, w):
    from eprecon_amd.backbone import dwconv_nhwc
    torch.manual_seed(k * 10 + s)
    conv = nn.Conv2d(c, c, k, padding=k // 2, stride=s, groups=c, bias=False).cuda()
    x = torch.randn(3, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = dwconv_nhwc(conv, x, 3)
        ref = F.conv2d(x.double().cpu(), conv.weight.double().cpu(), None, s, k // 2, 1, c)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(y.cpu().double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b", [1, 2])
def test_depthwise_conv_applies_the_pending_batchnorm(b):
    from eprecon_amd.backbone import bn_views_stats, dwconv_nhwc
    torch.manual_seed(0)
    v, c, h, w = 3, 48, 20, 28
    conv = nn.Conv2d(c, c, 3, padding=1, stride=2, groups=c, bias=False).cuda()
    bn = nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        x = (torch.randn(v * b, c, h, w, device="cuda") * 3 + 1).contiguous(memory_format=torch.channels_last)
        y = dwconv_nhwc(conv, x, v, (bn_views_stats(bn, x, v), True))
        ref = torch.cat([conv(F.relu(F.batch_norm(x[i * b:(i + 1) * b], None, None, bn.weight, bn.bias, True, 0.0, bn.eps)))
                         for i in range(v)])
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4), float((y - ref).abs().max())


@pytest.mark.parametrize("c,h,w", [(16, 240, 320), (72, 60, 80), (480, 30, 40), (24, 7, 5)])
@pytest.mark.parametrize("relu,res", [(True, False), (False, True)])
def test_batchnorm_per_view(c, h, w, relu, res):
    from eprecon_amd.backbone import bn_views_apply, bn_views_stats
    torch.manual_seed(c)
    v = 3
    bn = nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        x = (torch.randn(v, c, h, w, device="cuda") * 2 + 5).contiguous(memory_format=torch.channels_last)   # mean >> 0
        r = torch.randn_like(x) if res else None
        ref = F.instance_norm(x.double(), weight=bn.weight.double(), bias=bn.bias.double(), use_input_stats=True, eps=bn.eps)
        ref = F.relu(ref) if relu else ref
        ref = ref + r.double() if res else ref
        aff = bn_views_stats(bn, x, v)
        aff2 = bn_views_stats(bn, x, v)
        y = bn_views_apply(x.clone(memory_format=torch.channels_last), aff, v, relu, r)
    assert torch.equal(aff, aff2)           # deterministic (the merge order is fixed, whoever merges)
    assert torch.allclose(y.double(), ref, rtol=2e-5, atol=2e-5), float((y.double() - ref).abs().max())


@pytest.mark.parametrize("b", [1, 2])
def test_forward_views_equals_per_view_calls(b, monkeypatch):
    """the whole trunk + FPN head: one batched pass on the HIP layers == V separate train-mode forward() calls"""
    import eprecon_amd.backbone as BB
    torch.manual_seed(3)
    net = BB.MnasMulti(1.0).cuda().train()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.7, 1.3)
                m.bias.uniform_(-0.2, 0.2)
    imgs = [torch.randn(b, 3, 96, 128, device="cuda") for _ in range(3)]
    with torch.no_grad():
        got = net.forward_views(imgs)
        monkeypatch.setattr(BB, "BACKBONE_HIP", False)
        torch_path = net.forward_views(imgs)
        ref = [net(im) for im in imgs]
    for vi in range(3):
        for lvl in range(3):
            scale = float(ref[vi][lvl].abs().max())
            assert torch.allclose(got[vi][lvl], ref[vi][lvl], rtol=2e-3, atol=2e-3 * scale), (vi, lvl, float((got[vi][lvl] - ref[vi][lvl]).abs().max()), scale)
            assert torch.allclose(torch_path[vi][lvl], ref[vi][lvl], rtol=2e-3, atol=2e-3 * scale)
