"""2D image feature extractor — mirror of models/backbone.py:22-77 (MnasMulti): MNASNet-1.0 trunk up
to stride 16 + FPN head -> [f4 (24ch, 1/4), f8 (40ch, 1/8), f16 (80ch, 1/16)].

This is the feeder of the 3D path and stays PyTorch-ROCm (MIOpen), as BASELINE.json prescribes.
torchvision (and its pretrained download) is not available in this environment, so the MNASNet
trunk is defined here with the layer layout and parameter names of torchvision's `MNASNet.layers`
([0..7] stem, [8] 16->24 k3 s2 e3 x3, [9] 24->40 k5 s2 e3 x3, [10] 40->80 k5 s2 e6 x3), which
keeps reference checkpoints loadable; weights are random-initialised.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


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
        c0 = self._run(self.conv0, x, v)
        c1 = self._run(self.conv1, c0, v)
        c2 = self._run(self.conv2, c1, v)
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
