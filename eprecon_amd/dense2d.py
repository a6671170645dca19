"""Dense 2D convolutions of the initialisation branch on the HIP gather-GEMM kernel.

A `nn.Conv2d(padding="same")` over V images of H x W pixels stored channels-last IS a sparse
convolution over the V*H*W pixel rows whose kernel map is known in closed form
(csrc/kernel_map.hip: pixel_map_kernel).  The 2D fusion stack of Occupancy_Initialization
(models/occupancy_initialization.py:22-58, models/modules.py:313-399 of the reference) therefore runs on the
same MFMA kernel as the sparse 3D layers, with its epilogues fused: bias, ReLU, the residual of
BN(x + ReLU(conv(x))) blocks, and the statistics pass of the train-mode BatchNorm that follows every
convolution (the reference tests in train mode, main.py:357).  Activations are [V*H*W, C] row-major
matrices (= channels-last tensors); layers write straight into channel slices of concat buffers.

The nn.Conv2d / nn.BatchNorm2d modules stay the parameter holders (state-dict compatible with the
reference); weights are re-laid out to [ky*k+kx][C_in][C_out] once per parameter version.
"""
import ctypes
import os

import torch

from . import _lib
from . import sparse as SP


class PixelGrid:
    """The V*H*W pixel rows of a stack of V channels-last maps, with the cached 3x3 kernel map."""
    _cache = {}

    def __init__(self, maps, height, width, device):
        self.maps, self.height, self.width, self.device = maps, height, width, device
        self.n = maps * height * width
        self._nbr = {}

    @classmethod
    def get(cls, maps, height, width, device):
        key = (maps, height, width, str(device))
        g = cls._cache.get(key)
        if g is None:
            g = cls._cache[key] = cls(maps, height, width, device)
        return g

    def kernel_map(self, ksize):
        if ksize == 1:
            return None
        nbr = self._nbr.get(ksize)
        if nbr is None:
            nbr = torch.empty((ksize * ksize, self.n), dtype=torch.int32, device=self.device)
            _lib.check(_lib.load().eprecon_pixel_map_async(self.maps, self.height, self.width, ksize,
                                                           _lib.ptr(nbr), _lib.current_stream()),
                       "eprecon_pixel_map_async")
            self._nbr[ksize] = nbr
        return nbr


def rows_of(x):
    """channels-last [V,C,H,W] tensor -> its [V*H*W, C] row-major view (no copy)"""
    v, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(v * h * w, c)


def maps_of(rows, maps, height, width):
    """[V*H*W, C] contiguous rows -> the channels-last [V,C,H,W] view"""
    return rows.view(maps, height, width, rows.shape[1]).permute(0, 3, 1, 2)


def packed_weight(conv):
    """nn.Conv2d weight [C_out, C_in, k, k] -> f32[k*k, C_in, C_out], cached per parameter version"""
    w = conv.weight
    tag = (w._version, w.data_ptr(), w.device)
    cached = getattr(conv, "_eprecon_packed", None)
    if cached is None or cached[0] != tag:
        co, ci, kh, kw = w.shape
        assert kh == kw and kh % 2 == 1 and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
        cached = (tag, w.detach().permute(2, 3, 1, 0).reshape(kh * kw, ci, co).contiguous())
        conv._eprecon_packed = cached
    return cached[1]


# 3x3 layers on long pixel lists through the direct gather kernel (csrc/sparse_conv_direct_impl.hpp) on the pixel map instead of the
# image-tile kernel: rocprofv3 durations on the 9 x 120 x 160 level 24->12 36.6 -> 23.4 us, 12->12 28.6 -> 17.6, 24->24 38.2 ->
# 36.8; on 9 x 60 x 80 it was a wash in round 4 (40->40 33.7 -> 28.7, 32->32 17.5 -> 18.4).  Round 6, with the direct kernel's loads
# spread among its MFMAs: the 43,200-pixel level on it too takes the cfg2 step from 1.491 to 1.457 ms (tools/probes/cfg2_direct2d.py,
# three interleaved pairs); the 10,800-pixel level stays on the split-K kernel (1.459 / 1.494 against 1.448 / 1.467: inside the noise).
DIRECT_2D_MIN_ROWS = 40000
K1_DIRECT_2D_MIN_ROWS = 20000     # point-wise layers of the stack (1 x 1 convolutions): see conv_bn_launch


# EPRECON_BN_ACC=1: the BatchNorms of the 2D fusion stack finished by their CONSUMERS from order-independent integer accumulators
# (round 6, csrc/conv_common.hpp "BatchNorm form (c)") instead of by a finalize launch behind every layer.  Built, parity-green,
# deterministic — and SLOWER on this part, hence opt-in: cfg2 1.823 ms against 1.600 (profiles/r06/bn_acc_ab.txt).  Device-scope
# atomics cost 37 us per step for every int64 word a workgroup adds per channel (seven words here: +0.26 ms), and even with
# the atomics compiled out (wrong results, timing only) the step takes 1.56 ms: the 37 finalize launches it removes were worth
# 0.04 ms, not the 0.2 ms their summed durations suggest (three branches of the stack run side by side).
BN_ACC = os.environ.get("EPRECON_BN_ACC", "0") == "1"


class BnArena:
    """The accumulator blocks of ONE pass over a stack of layers: carved out of one int64 buffer that a single fill zeroes
    at the start of the pass (`begin`).  A block must be zero before its producer runs and is read by every consumer of the
    layer, so blocks are never reused inside a pass; the next pass reuses the same addresses (HIP-graph replays included:
    the fill is part of the captured pass)."""

    def __init__(self, device, words=1 << 19):
        self.buf = torch.zeros(words, dtype=torch.int64, device=device)
        self.offset = 0
        self.high = words          # extent to clear at the next begin(): everything the first time

    def begin(self):
        if self.high:
            self.buf[:self.high].zero_()
        self.offset = 0
        self.high = 0

    def take(self, words):
        words = (words + 1) & ~1          # 16-byte aligned blocks
        if self.offset + words > self.buf.numel():
            raise _lib.EpreconError("BatchNorm accumulator arena exhausted (eprecon_amd.dense2d.BnArena)")
        t = self.buf[self.offset:self.offset + words]
        self.offset += words
        self.high = max(self.high, self.offset)
        return t


_ARENA = None     # the arena of the pass being issued (set by `bn_pass`); None: BatchNorms are finished by finalize launches


class bn_pass:
    """with bn_pass(arena): the layers issued inside leave their BatchNorms in accumulator blocks of `arena` (zeroed here)"""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        global _ARENA
        self.prev, _ARENA = _ARENA, self.arena
        if self.arena is not None:
            self.arena.begin()
        return self.arena

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = self.prev
        return False


class AccSlice:
    """channels [c0, c0 + c) of an accumulator block of `ld` channels (`block`: the int64 tensor)"""
    __slots__ = ("block", "ld", "c0", "c")

    def __init__(self, block, ld, c0, c):
        self.block, self.ld, self.c0, self.c = block, ld, c0, c

    @classmethod
    def new(cls, channels):
        words = int(_lib.load().eprecon_bn_acc_words(channels))
        return cls(_ARENA.take(words), channels, 0, channels)

    def part(self, a, b):
        return AccSlice(self.block, self.ld, self.c0 + a, b - a)


class Act:
    """An activation of the 2D stack: rows f32[N, C] (possibly a channel slice of a concat buffer) plus
    the BatchNorm of its producer still pending — in affine form (scale, shift f32[C], ReLU flag) or, round 6, as the
    accumulator block its producer summed into (acc: AccSlice, eps).  The stored rows are the un-normalised convolution
    output; consumers apply the BatchNorm while gathering.  scale and acc are None for a materialised tensor."""
    __slots__ = ("rows", "scale", "shift", "relu", "acc", "eps")

    def __init__(self, rows, scale=None, shift=None, relu=False, acc=None, eps=1e-5):
        self.rows, self.scale, self.shift, self.relu, self.acc, self.eps = rows, scale, shift, relu, acc, eps

    def affine(self):
        """(scale, shift) vectors of a pending BatchNorm held as accumulators: one launch (consumers that cannot finish the
        block themselves: residual operands)"""
        if self.acc is not None and self.scale is None:
            a = torch.empty((2, self.acc.c), dtype=torch.float32, device=self.rows.device)
            _lib.check(_lib.load().eprecon_batchnorm_acc_affine_async(
                self.acc.block.data_ptr(), self.acc.ld, self.acc.c0, self.acc.c, float(self.eps), a[0].data_ptr(), a[1].data_ptr(),
                _lib.current_stream()), "eprecon_batchnorm_acc_affine_async")
            self.scale, self.shift = a[0], a[1]
        return self.scale, self.shift


def _dptr(t):
    return None if t is None else t.data_ptr()


def conv_bn_act(conv, bn, x, grid, out=None, aff=None, relu=True, pre_relu=False, residual=None):
    """One launch (+ a tiny finalize) for  BN( [ReLU](conv(x) + b) [+ residual] )  with x / residual given
    as Acts (their pending BatchNorms are applied on load) and the BatchNorm of the result left pending:
    returns Act(raw rows, scale, shift, relu).  `out`: rows to write (may be a channel slice), `aff`: the
    (scale, shift) slices to fill (e.g. of a concat buffer's vectors)."""
    return conv_bn_launch(packed_weight(conv), conv.bias, bn.weight, bn.bias, bn.eps, conv.kernel_size[0], x, grid,
                          out=out, aff=aff, relu=relu, pre_relu=pre_relu, residual=residual)


def conv_bn_launch(w, bias, gamma, beta, eps, k, x, grid, out=None, aff=None, relu=True, pre_relu=False,
                   residual=None):
    lib = _lib.load()
    kvol, cin, cout = w.shape
    rows = x.rows
    n = rows.shape[0]
    dev = rows.device
    assert rows.shape[1] == cin and rows.dtype == torch.float32 and rows.stride(1) == 1
    nbr = grid.kernel_map(k)
    if out is None:
        out = torch.empty((n, cout), dtype=torch.float32, device=dev)
    d = _lib.ConvDesc()
    d.x, d.n_in, d.ld_x = rows.data_ptr(), n, rows.stride(0)
    d.nbr, d.kvol, d.n_out = _dptr(nbr), kvol, n
    d.weight, d.cin, d.cout = w.data_ptr(), cin, cout
    d.bias = _dptr(bias)
    if residual is not None:
        assert residual.rows.shape == (n, cout)
        d.residual, d.ld_res = residual.rows.data_ptr(), residual.rows.stride(0)
        r_scale, r_shift = residual.affine()
        d.res_scale, d.res_shift, d.res_relu = _dptr(r_scale), _dptr(r_shift), int(residual.relu)
    d.out, d.ld_out = out.data_ptr(), out.stride(0)
    d.relu, d.accumulate = int(pre_relu), 0
    if x.acc is not None and x.scale is None:     # the input's BatchNorm is finished by this launch's prologue
        assert x.acc.c == cin
        d.in_acc, d.in_acc_ld, d.in_acc_c0, d.in_eps = x.acc.block.data_ptr(), x.acc.ld, x.acc.c0, float(x.eps)
    else:
        d.in_scale, d.in_shift = _dptr(x.scale), _dptr(x.shift)
    d.in_relu = int(x.relu)
    if k == 3:
        d.img_h, d.img_w, d.img_maps = grid.height, grid.width, grid.maps  # narrow layers: image-tile kernel
        if n < DIRECT_2D_MIN_ROWS:
            pq = SP.packed_weight(w)                 # short pixel lists (the 10,800-pixel level): B operands of the split-K kernel
            d.packed_weight = pq.data_ptr()
        if cout <= SP.DIRECT_MAX_COUT and n >= DIRECT_2D_MIN_ROWS:
            pw = SP.packed_weight16(w)               # long pixel lists: the direct gather kernel on the pixel map
            d.packed_weight16 = pw.data_ptr()
    elif k == 1 and cout <= SP.DIRECT_MAX_COUT and n >= K1_DIRECT_2D_MIN_ROWS:
        # point-wise layers on long pixel lists: the direct kernel as a streaming product (sparse.K1_DIRECT_MIN_ROWS' rule; the
        # slab kernel took 144 -> 32 on 43,200 pixels in 23 us, 160 -> 40 in 34, 96 -> 24 on 172,800 in 37: 14 / 19 / 30 on this one)
        pw = SP.packed_weight16(w)
        d.packed_weight16 = pw.data_ptr()
    if _ARENA is not None and (aff is None or isinstance(aff, AccSlice)) and lib.eprecon_conv_desc_takes_bn_acc(ctypes.byref(d)):
        # the launch sums into an accumulator block; whoever consumes the result finishes the BatchNorm: no finalize launch
        acc = aff if aff is not None else AccSlice.new(cout)
        assert acc.c == cout
        d.bn_acc, d.bn_acc_ld, d.bn_acc_c0 = acc.block.data_ptr(), acc.ld, acc.c0
        d.bn_gamma, d.bn_beta = _dptr(gamma), _dptr(beta)
        _lib.check(lib.eprecon_conv_desc_async(ctypes.byref(d), _lib.current_stream()), "eprecon_conv_desc_async")
        return Act(out, relu=relu, acc=acc, eps=eps)
    if aff is None:
        a = torch.empty((2, cout), dtype=torch.float32, device=dev)
        aff = (a[0], a[1])
    assert not isinstance(aff, AccSlice), "an accumulator slice was handed to a launch that cannot produce into it"
    # the summaries are per workgroup: 128-row blocks (gather forms) or image tiles (tile kernel)
    partial = torch.empty((lib.eprecon_conv_desc_partial_rows(ctypes.byref(d)), 3, cout), dtype=torch.float32, device=dev)
    d.bn_partial = partial.data_ptr()
    _lib.check(lib.eprecon_conv_desc_async(ctypes.byref(d), _lib.current_stream()), "eprecon_conv_desc_async")
    _lib.check(lib.eprecon_batchnorm_finalize_affine_async(
        partial.data_ptr(), partial.shape[0], cout, _dptr(gamma), _dptr(beta), float(eps),
        aff[0].data_ptr(), aff[1].data_ptr(), _lib.current_stream()), "eprecon_batchnorm_finalize_affine_async")
    return Act(out, aff[0], aff[1], relu)


def materialize(act, out=None):
    """apply the pending BatchNorm (+ReLU): returns plain rows"""
    if act.scale is None and act.acc is None:
        if out is not None and out.data_ptr() != act.rows.data_ptr():
            out.copy_(act.rows)
            return out
        return act.rows
    rows = act.rows
    n, c = rows.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=rows.device)
    if act.scale is None and c <= 512:      # every workgroup finishes the accumulator block itself: one launch
        _lib.check(_lib.load().eprecon_affine_rows_acc_async(
            rows.data_ptr(), n, c, rows.stride(0), act.acc.block.data_ptr(), act.acc.ld, act.acc.c0, float(act.eps), int(act.relu),
            out.data_ptr(), out.stride(0), _lib.current_stream()), "eprecon_affine_rows_acc_async")
        return out
    act.affine()
    _lib.check(_lib.load().eprecon_affine_rows_async(
        rows.data_ptr(), n, c, rows.stride(0), act.scale.data_ptr(), act.shift.data_ptr(), int(act.relu),
        out.data_ptr(), out.stride(0), _lib.current_stream()), "eprecon_affine_rows_async")
    return out


def conv_bn(conv, bn, x, grid, out=None, relu=True, pre_relu=False, pre_residual=None):
    """BN( [ReLU](conv(x) + b) [+ pre_residual] ) [ReLU] on plain pixel rows, result materialised"""
    res = Act(pre_residual) if pre_residual is not None else None
    return materialize(conv_bn_act(conv, bn, Act(x), grid, relu=relu, pre_relu=pre_relu, residual=res), out=out)
