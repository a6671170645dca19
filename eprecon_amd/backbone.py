"""2D image feature extractor — mirror of models/backbone.py:22-77 (MnasMulti): MNASNet-1.0 trunk up
to stride 16 + FPN head -> [f4 (24ch, 1/4), f8 (40ch, 1/8), f16 (80ch, 1/16)].

This is the feeder of the 3D path.  The point-wise and dense 3 x 3 convolutions stay PyTorch-ROCm (hipBLASLt / MIOpen), as
BASELINE.json prescribes; on the batched GPU inference path (forward_views) the depthwise convolutions and the per-view
train-mode BatchNorms run on csrc/backbone2d.hip (MIOpen serves fp32 channels-last depthwise layers with its naive
reference kernel: 1.85 of the 4.4 ms of a pass).  EPRECON_BACKBONE_HIP=0: everything on PyTorch ops.
torchvision (and its pretrained download) is not available in this environment, so the MNASNet
trunk is defined here with the layer layout and parameter names of torchvision's `MNASNet.layers`
([0..7] stem, [8] 16->24 k3 s2 e3 x3, [9] 24->40 k5 s2 e3 x3, [10] 40->80 k5 s2 e6 x3), which
keeps reference checkpoints loadable; weights are random-initialised.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

BACKBONE_HIP = os.environ.get("EPRECON_BACKBONE_HIP", "1") == "1"


def _round_to_multiple_of(val, divisor, round_up_bias=0.9):
    new_val = max(divisor, int(val + divisor / 2) // divisor * divisor)
    return new_val if new_val >= round_up_bias * val else new_val + divisor


def _get_depths(alpha):
    return [_round_to_multiple_of(d * alpha, 8) for d in (32, 16, 24, 40, 80, 96, 192, 320)]


class _InvertedResidual(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride, expansion):
        super().__init__()
        mid = in_ch * expansion
        self.apply_residual = in_ch == out_ch and stride == 1
        self.layers = nn.Sequential(
            nn.Conv2d(in_ch, mid, 1, bias=False), nn.BatchNorm2d(mid), nn.ReLU(inplace=True),
            nn.Conv2d(mid, mid, kernel_size, padding=kernel_size // 2, stride=stride, groups=mid, bias=False),
            nn.BatchNorm2d(mid), nn.ReLU(inplace=True),
            nn.Conv2d(mid, out_ch, 1, bias=False), nn.BatchNorm2d(out_ch))

    def forward(self, x):
        y = self.layers(x)
        return y + x if self.apply_residual else y


def _stack(in_ch, out_ch, kernel_size, stride, expansion, repeats):
    blocks = [_InvertedResidual(in_ch, out_ch, kernel_size, stride, expansion)]
    blocks += [_InvertedResidual(out_ch, out_ch, kernel_size, 1, expansion) for _ in range(repeats - 1)]
    return nn.Sequential(*blocks)


class MnasMulti(nn.Module):
    def __init__(self, alpha=1.0):
        super().__init__()
        d = _get_depths(alpha)
        self.conv0 = nn.Sequential(
            nn.Conv2d(3, d[0], 3, padding=1, stride=2, bias=False), nn.BatchNorm2d(d[0]), nn.ReLU(inplace=True),
            nn.Conv2d(d[0], d[0], 3, padding=1, stride=1, groups=d[0], bias=False), nn.BatchNorm2d(d[0]),
            nn.ReLU(inplace=True),
            nn.Conv2d(d[0], d[1], 1, bias=False), nn.BatchNorm2d(d[1]),
            _stack(d[1], d[2], 3, 2, 3, 3))
        self.conv1 = _stack(d[2], d[3], 5, 2, 3, 3)
        self.conv2 = _stack(d[3], d[4], 5, 2, 6, 3)
        self.out1 = nn.Conv2d(d[4], d[4], 1, bias=False)
        self.inner1 = nn.Conv2d(d[3], d[4], 1, bias=True)
        self.inner2 = nn.Conv2d(d[2], d[4], 1, bias=True)
        self.out2 = nn.Conv2d(d[4], d[3], 3, padding=1, bias=False)
        self.out3 = nn.Conv2d(d[4], d[2], 3, padding=1, bias=False)
        self.out_channels = [d[4], d[3], d[2]]
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        return self._head(c0, c1, c2)

    def _head(self, c0, c1, c2):
        f16 = self.out1(c2)
        top = F.interpolate(c2, scale_factor=2, mode="nearest") + self.inner1(c1)
        f8 = self.out2(top)
        top = F.interpolate(top, scale_factor=2, mode="nearest") + self.inner2(c0)
        f4 = self.out3(top)
        return [f4, f8, f16]

    # ---- all views of a fragment in ONE pass --------------------------------------------
    def forward_views(self, imgs):
        """imgs: list of V tensors [B,3,H,W] (what models/neuralrecon.py:49 unbinds) -> list over views of
        [f4, f8, f16], the same values as V separate forward() calls, computed as one [V*B] batch in
        channels-last memory.  The reference runs the backbone in TRAIN mode at test time (main.py:357), so every
        BatchNorm normalises with the statistics of the call's own batch — one view's B images; batching the views
        must not mix them, hence BatchNorm is evaluated per view (`_bn_per_view`).  The returned maps are slices of
        one channels-last tensor per level: stacked over views they are consumed by the back-projection in place
        (no NCHW -> NHWC re-layout, no copy)."""
        v, b = len(imgs), imgs[0].shape[0]
        x = torch.cat(list(imgs), 0).contiguous(memory_format=torch.channels_last)
        run = self._run
        if BACKBONE_HIP and x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.float32:
            run = self._run_hip
        c0 = run(self.conv0, x, v)
        c1 = run(self.conv1, c0, v)
        c2 = run(self.conv2, c1, v)
        levels = self._head(c0, c1, c2)
        return [[lvl[i * b:(i + 1) * b] for lvl in levels] for i in range(v)]

    def _run(self, mod, x, v):
        if isinstance(mod, nn.BatchNorm2d):
            return _bn_per_view(mod, x, v)
        if isinstance(mod, _InvertedResidual):
            y = self._run(mod.layers, x, v)
            return y + x if mod.apply_residual else y
        if isinstance(mod, nn.Sequential):
            for m in mod:
                x = self._run(m, x, v)
            return x
        return mod(x)


    # ---- the same on csrc/backbone2d.hip: depthwise convolutions + per-view BatchNorm ----
    def _run_hip(self, mod, x, v, residual=None):
        """a Sequential walked with one module of look-ahead: a BatchNorm (+ ReLU) in front of a depthwise convolution
        stays PENDING and is applied by that convolution on load; any other BatchNorm (+ ReLU) is statistics + one apply
        pass, which also adds the block's skip (`residual`, at the last BatchNorm of an inverted-residual block)"""
        if isinstance(mod, _InvertedResidual):
            return self._run_hip(mod.layers, x, v, residual=x if mod.apply_residual else None)
        if not isinstance(mod, nn.Sequential):
            return mod(x)
        mods = list(mod)
        i, pending = 0, None
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.BatchNorm2d):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                nxt = mods[i + 1 + int(relu)] if i + 1 + int(relu) < len(mods) else None
                aff = bn_views_stats(m, x, v)
                if _is_depthwise(nxt) and _hip_ok(x):
                    pending = (aff, relu)
                else:
                    last = i + 1 + int(relu) >= len(mods)
                    x = bn_views_apply(x, aff, v, relu, residual if last else None)
                    if last:
                        residual = None
                i += 1 + int(relu)
            elif _is_depthwise(m) and _hip_ok(x):
                x = dwconv_nhwc(m, x, v, pending)
                pending = None
                i += 1
            elif isinstance(m, (nn.Sequential, _InvertedResidual)):
                x = self._run_hip(m, x, v)
                i += 1
            else:
                assert pending is None
                x = m(x)
                i += 1
        assert pending is None
        return x if residual is None else x + residual


def _is_depthwise(m):
    return (isinstance(m, nn.Conv2d) and m.groups == m.in_channels == m.out_channels and m.groups > 1 and m.bias is None
            and m.kernel_size in ((3, 3), (5, 5)) and m.stride in ((1, 1), (2, 2)) and m.padding == (m.kernel_size[0] // 2,) * 2
            and m.dilation == (1, 1) and m.in_channels % 4 == 0)


def _hip_ok(x):
    return x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.shape[1] <= 480 \
        and x.is_contiguous(memory_format=torch.channels_last)


def bn_views_stats(bn, x, v):
    """train-mode BatchNorm2d of a channels-last [V*B, C, H, W] batch with separate statistics per view, as
    (scale, shift) f32[V, 2, C] (eprecon_bn2d_views_stats_async); PyTorch ops when the map is not taken by the kernel"""
    n, c, h, w = x.shape
    rows = (n // v) * h * w
    if not _hip_ok(x) or v > 256:
        x5 = x.reshape(v, n // v, c, h * w).transpose(1, 2).reshape(v, c, -1)
        var, mean = torch.var_mean(x5, dim=2, unbiased=False)
        sc = bn.weight / torch.sqrt(var + bn.eps)
        return torch.stack([sc, bn.bias - mean * sc], 1).contiguous()
    lib = _lib.load()
    aff = torch.empty((v, 2, c), dtype=torch.float32, device=x.device)
    ws = _lib.workspace(lib.eprecon_bn2d_views_workspace_bytes(v, rows, c), x.device)
    _lib.check(lib.eprecon_bn2d_views_stats_async(_lib.ptr(x), v, rows, c, _lib.ptr(bn.weight), _lib.ptr(bn.bias), float(bn.eps),
                                                  _lib.ptr(aff), _lib.ptr(ws), ws.numel(),
                                                  _lib.current_stream()), "eprecon_bn2d_views_stats_async")
    return aff


def bn_views_apply(x, aff, v, relu, residual=None):
    """[relu](x * scale + shift) [+ residual] per view, in place on a channels-last map"""
    n, c, h, w = x.shape
    if not _hip_ok(x) or (residual is not None and not _hip_ok(residual)):
        b = n // v
        sc = aff[:, 0].repeat_interleave(b, 0)[:, :, None, None]
        sh = aff[:, 1].repeat_interleave(b, 0)[:, :, None, None]
        y = x * sc + sh
        y = F.relu(y) if relu else y
        return y if residual is None else y + residual
    _lib.check(_lib.load().eprecon_bn2d_views_apply_async(_lib.ptr(x), v, (n // v) * h * w, c, _lib.ptr(aff), int(relu),
                                                          _lib.ptr(residual), _lib.ptr(x), _lib.current_stream()),
               "eprecon_bn2d_views_apply_async")
    return x


def _dw_taps(conv):
    """[C, 1, k, k] -> tap-major [k*k, C], cached per weight version"""
    w = conv.weight
    tag = (w._version, w.data_ptr())
    hit = getattr(conv, "_eprecon_packed", None)
    if hit is None or hit[0] != tag:
        hit = (tag, w.detach().reshape(w.shape[0], -1).t().contiguous())
        conv._eprecon_packed = hit
    return hit[1]


def dwconv_nhwc(conv, x, v, pending=None):
    """depthwise conv of a channels-last map on eprecon_dwconv2d_nhwc_async; pending = ((scale, shift) f32[V,2,C], relu):
    the producer's BatchNorm (+ ReLU), applied on load"""
    n, c, h, w = x.shape
    k, s = conv.kernel_size[0], conv.stride[0]
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    aff, relu = pending if pending is not None else (None, False)
    _lib.check(_lib.load().eprecon_dwconv2d_nhwc_async(_lib.ptr(x), n, h, w, c, _lib.ptr(_dw_taps(conv)), k, s, _lib.ptr(aff),
                                                       max(n // v, 1), int(relu), _lib.ptr(out), _lib.current_stream()),
               "eprecon_dwconv2d_nhwc_async")
    return out


def _bn_per_view(bn, x, v):
    """train-mode BatchNorm2d of a [V*B, C, H, W] batch with SEPARATE statistics per view (B images each): for
    B = 1 that is an instance norm with the BatchNorm's affine parameters (one fused kernel); for B > 1 the
    (B, H, W) axes of a view are folded into the instance's spatial extent.  Running statistics are not
    updated (they are never read: the reference never leaves train mode)."""
    n, c, h, w = x.shape
    b = n // v
    if b == 1:
        return F.instance_norm(x, weight=bn.weight, bias=bn.bias, use_input_stats=True, eps=bn.eps)
    x5 = x.reshape(v, b, c, h, w).transpose(1, 2)                      # [V, C, B, H, W]
    y = F.instance_norm(x5.reshape(v, c, b * h, w), weight=bn.weight, bias=bn.bias, use_input_stats=True, eps=bn.eps)
    return y.reshape(v, c, b, h, w).transpose(1, 2).reshape(n, c, h, w).contiguous(memory_format=torch.channels_last)


def stack_views(maps):
    """torch.stack(maps) for V tensors [B,C,h,w] WITHOUT a copy when they are consecutive slices of one tensor
    (forward_views): the result aliases that tensor as [V,B,C,h,w]."""
    m0 = maps[0]
    if torch.is_grad_enabled() and any(m.requires_grad for m in maps):
        return torch.stack(list(maps))      # training: a recorded copy instead of an alias
    step = m0.shape[0] * m0.stride(0)
    base = m0.untyped_storage().data_ptr()
    if step > 0 and all(m.shape == m0.shape and m.stride() == m0.stride() and m.dtype == m0.dtype
                        and m.untyped_storage().data_ptr() == base      # slices of ONE allocation, not neighbours
                        and m.storage_offset() == m0.storage_offset() + i * step for i, m in enumerate(maps)):
        return torch.as_strided(m0, (len(maps),) + tuple(m0.shape), (step,) + tuple(m0.stride()))
    return torch.stack(list(maps))
