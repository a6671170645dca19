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


def conv_bn(conv, bn, x, grid, out=None, relu=True, pre_relu=False, pre_residual=None):
    """BN( [ReLU](conv(x) + b) [+ pre_residual] ) [ReLU] on pixel rows: one fused convolution launch
    (bias, ReLU, residual, BatchNorm summaries in the epilogue) + finalize + apply.  `out` may be a
    channel slice of a wider buffer; returns it."""
    w = packed_weight(conv)
    k = conv.kernel_size[0]
    y, partial = SP.sparse_conv_fused(x, w if k > 1 else w[0], grid.kernel_map(k), conv.bias, out=out,
                                      relu=pre_relu, residual=pre_residual, bn_partial=True)
    return SP.batchnorm_apply_partials(y, partial, bn.weight, bn.bias, bn.eps, relu=relu, out=y)
