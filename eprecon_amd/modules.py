"""Building blocks of the 3D path — host-side mirror of models/modules.py of the reference.

Two families:
  * sparse layers, executed by libeprecon_hip.so through eprecon_amd.sparse:
      SparseSubMConv3d, SubMconv3dBlock, Spares3dELAN, SparseConv3d_Residual     (spconv in the reference)
      SPVCNN, SConv3d, ConvGRU and their blocks                                  (torchsparse in the reference)
  * small dense blocks that stay PyTorch-ROCm (MIOpen / rocBLAS), as BASELINE.json's north_star
    prescribes for the 2D side and the heads: Conv2d_Block, Conv2d_Residual_Block, ELAN,
    Fusion_Block, Linear4xTrans, Linear_Residual.

Class names, constructor arguments, forward signatures and parameter names follow the reference so
that its state_dict keys line up, except the sparse conv weights, whose layout is this build's
[K^3, C_in, C_out] with x-fastest offset order (spconv stores [C_out, kx, ky, kz, C_in];
`SparseSubMConv3d.load_spconv_weight` converts).

Every sparse module has two wirings over the same parameters: the fused inference launches (in-place concat buffers,
pending BatchNorms applied on load, gate epilogues) under torch.no_grad(), and, with autograd enabled, the recording
operators of eprecon_amd/autograd.py whose backward runs on the HIP kernels as well (`recording()`).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as AG
from . import dense2d as D2
from . import sparse as SP


def recording():
    """True on the training path: the recording operators of eprecon_amd/autograd.py run instead of the fused
    inference launches (which write in place and keep no graph).  The reference's inference loop runs under
    torch.no_grad() (main.py:351-365), and so does every inference entry point of this package."""
    return torch.is_grad_enabled()

# ------------------------------------------------------------------------------------------------
# coordinate-set cache: the reference hands raw `coords` tensors to every layer; spconv rebuilds
# its indice pairs each time (indice_key=None).  Here the hash grid + 27-neighbour table is built
# once per distinct coords tensor and shared by all layers that see the same tensor.
# ------------------------------------------------------------------------------------------------
# The cache is keyed on the tensor OBJECT (identity + data_ptr + torch's version counter).  Writes that the HIP library
# makes through raw pointers do not bump that counter: a caller that refills the SAME coords tensor in place (e.g. a
# persistent serving buffer) must call clear_voxel_set_cache() (and torchsparse_utils.clear_voxelization_cache()) at the
# fragment boundary, or hand over fresh tensors per fragment as every path in this package does.
_SET_CACHE = []
_SET_CACHE_MAX = 8
_SET_CACHE_LOCK = __import__("threading").Lock()   # the pipelined serving mode calls in from a worker thread as well


def voxel_set_for(coords, stride=1):
    key = (coords.data_ptr(), coords.shape[0], coords._version, int(stride), coords.device)
    with _SET_CACHE_LOCK:
        for k, ref, vs in _SET_CACHE:
            if k == key and ref is coords:
                return vs
    c = coords if coords.dtype == torch.int32 else coords.to(torch.int32)
    vs = SP.VoxelSet(c.contiguous(), stride)
    with _SET_CACHE_LOCK:
        _SET_CACHE.append((key, coords, vs))
        if len(_SET_CACHE) > _SET_CACHE_MAX:
            _SET_CACHE.pop(0)
    return vs


def clear_voxel_set_cache():
    with _SET_CACHE_LOCK:
        _SET_CACHE.clear()


# ------------------------------------------------------------------------------------------------
# spconv-style submanifold layers
# ------------------------------------------------------------------------------------------------
class SparseSubMConv3d(nn.Module):
    """spconv.SubMConv3d(C_in, C_out, Kernel, bias=True) — models/modules.py:249-271.
    Output voxels == input voxels; y_i = b + sum_o W_o^T x_j over active neighbours j."""

    def __init__(self, C_in, C_out, Kernel, Stride=1):
        super().__init__()
        assert Kernel in (1, 3) and Stride == 1
        self.kernel = Kernel
        kvol = Kernel ** 3
        self.weight = nn.Parameter(torch.empty(kvol, C_in, C_out))
        self.bias = nn.Parameter(torch.zeros(C_out))
        self.init_weights()

    def init_weights(self):
        # the reference applies xavier_uniform_ to spconv's [C_out, k, k, k, C_in] tensor
        # (models/modules.py:256-258): fan_in = k * (k*k*C_in), fan_out = C_out * (k*k*C_in)
        k = self.kernel
        _, cin, cout = self.weight.shape
        bound = math.sqrt(6.0 / (k * k * k * cin + cout * k * k * cin))
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            self.bias.zero_()

    @staticmethod
    def from_spconv_layout(w):
        """spconv 2.x stores SubMConv3d weights as [C_out, k0, k1, k2, C_in] with k0 acting on the FIRST
        spatial column of `indices`; the reference passes (b, x, y, z) rows (models/modules.py:267), so
        k0 = x.  This build's layout is [K^3, C_in, C_out] with flat offset k = (dz*3 + dy)*3 + dx
        (x fastest, csrc/kernel_map.hip), i.e. weight[k] = w[:, dx, dy, dz, :]^T."""
        co, k0, k1, k2, ci = w.shape
        return w.permute(3, 2, 1, 4, 0).reshape(k0 * k1 * k2, ci, co)

    def load_spconv_weight(self, w):
        with torch.no_grad():
            self.weight.copy_(self.from_spconv_layout(w))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """accepts the reference's checkpoints: `<name>.sparsesubmconv3d.{weight,bias}`
        (SparseSubMConv3d, models/modules.py:252) or a 5-D spconv-layout `<name>.weight`
        (SubMconv3dBlock.conv, models/modules.py:444)"""
        for leaf in ("weight", "bias"):
            old = prefix + "sparsesubmconv3d." + leaf
            if old in state_dict:
                state_dict[prefix + leaf] = state_dict.pop(old)
        w = state_dict.get(prefix + "weight")
        if w is not None and w.dim() == 5:
            state_dict[prefix + "weight"] = self.from_spconv_layout(w)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def run(self, features, vset, out=None, relu=False):
        if recording():
            nbr = vset.kernel_map(3) if self.kernel == 3 else None
            y = AG.sparse_conv(features, self.weight, nbr, self.bias)
            y = F.relu(y) if relu else y
            return y if out is None else out.copy_(y)
        nbr = vset.conv_map(3) if self.kernel == 3 else None   # dense-grid form on well-filled grids, else the kernel map
        return SP.sparse_conv(features, self.weight, nbr, self.bias, out=out, relu=relu)

    def run_stats(self, features, vset, out=None):
        """conv + bias and the BatchNorm summaries of its output in one launch -> (y, partial)"""
        nbr = vset.conv_map(3) if self.kernel == 3 else None
        return SP.conv_stats(features, self.weight, nbr, out=out, bias=self.bias)

    def run_ln(self, features, vset, ln, out=None, relu=False, residual=None, post_relu=False):
        """conv [+ReLU] [+residual] -> LayerNorm `ln` [-> ReLU], one launch"""
        if recording():
            nbr = vset.kernel_map(3) if self.kernel == 3 else None
            y = AG.sparse_conv(features, self.weight, nbr, self.bias)
            y = F.relu(y) if relu else y
            y = y + residual if residual is not None else y
            y = F.layer_norm(y, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
            y = F.relu(y) if post_relu else y
            return y if out is None else out.copy_(y)
        nbr = vset.conv_map(3) if self.kernel == 3 else None
        return SP.sparse_conv_ln(features, self.weight, nbr, self.bias, ln.weight, ln.bias, ln.eps, out=out,
                                 relu=relu, residual=residual, post_relu=post_relu)

    def forward(self, features, coords, spitial_shape, bs):
        """features f32[N, C_in]; coords int[N, 4] (b,x,y,z) -> f32[N, C_out]"""
        return self.run(features.contiguous(), voxel_set_for(coords))


class _RowLayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters, evaluated by the fused row-wise HIP epilogue"""

    def run(self, x, residual=None, pre_relu=False, post_relu=False, out=None):
        if recording():
            y = F.relu(x) if pre_relu else x
            y = y + residual if residual is not None else y
            y = F.layer_norm(y, self.normalized_shape, self.weight, self.bias, self.eps)
            y = F.relu(y) if post_relu else y
            return y if out is None else out.copy_(y)
        return SP.rowwise_layernorm(x, self.weight, self.bias, self.eps, residual, pre_relu, post_relu, out)


class SubMconv3dBlock(nn.Module):
    """conv -> LayerNorm -> ReLU (models/modules.py:440-452)"""

    def __init__(self, C_in, C_out, Kernel, Stride, Padding):
        super().__init__()
        self.conv = SparseSubMConv3d(C_in, C_out, Kernel, Stride)
        self.ln = _RowLayerNorm(C_out)

    def run(self, x, vset, out=None):
        return self.conv.run_ln(x, vset, self.ln, out=out, post_relu=True)


class Spares3dELAN(nn.Module):
    """ELAN over sparse voxels (models/modules.py:401-438): two 1x1x1 branches, a chain of four
    3x3x3 blocks at dim/2, channel concat (4*dim), 1x1x1 fusion back to dim."""

    def __init__(self, dim):
        super().__init__()
        h = dim // 2
        self.conv1 = SubMconv3dBlock(dim, dim, 1, 1, 0)
        self.conv2 = SubMconv3dBlock(dim, dim, 1, 1, 0)
        self.conv3 = SubMconv3dBlock(dim, h, 3, 1, 1)
        self.conv4 = SubMconv3dBlock(h, h, 3, 1, 1)
        self.conv5 = SubMconv3dBlock(h, h, 3, 1, 1)
        self.conv6 = SubMconv3dBlock(h, h, 3, 1, 1)
        self.conv7 = SubMconv3dBlock(dim * 4, dim, 1, 1, 0)
        self.dim = dim

    def run(self, x, vset):
        d, h = self.dim, self.dim // 2
        n = x.shape[0]
        if recording():
            x1, x2 = self.conv1.run(x, vset), self.conv2.run(x, vset)
            x3 = self.conv3.run(x2, vset)
            x4 = self.conv4.run(x3, vset)
            x5 = self.conv5.run(x4, vset)
            x6 = self.conv6.run(x5, vset)
            return self.conv7.run(torch.cat([x1, x2, x3, x4, x5, x6], dim=1), vset)
        # the concat buffer is written in place by each branch (no torch.cat copies)
        cat = torch.empty((n, 4 * d), dtype=torch.float32, device=x.device)
        self.conv1.run(x, vset, out=cat[:, 0:d])
        self.conv2.run(x, vset, out=cat[:, d:2 * d])
        self.conv3.run(cat[:, d:2 * d], vset, out=cat[:, 2 * d:2 * d + h])
        self.conv4.run(cat[:, 2 * d:2 * d + h], vset, out=cat[:, 2 * d + h:3 * d])
        self.conv5.run(cat[:, 2 * d + h:3 * d], vset, out=cat[:, 3 * d:3 * d + h])
        self.conv6.run(cat[:, 3 * d:3 * d + h], vset, out=cat[:, 3 * d + h:4 * d])
        return self.conv7.run(cat, vset)

    def forward(self, voxel_features_o, voxel_coords_bxyz, batch_size, spitial_shape):
        return self.run(voxel_features_o.contiguous(), voxel_set_for(voxel_coords_bxyz))


class SparseConv3d_Residual(nn.Module):
    """LN(x + ReLU(SubM(x))) — models/modules.py:469-482"""

    def __init__(self, dim, Kernel):
        super().__init__()
        self.SConv3d = SparseSubMConv3d(dim, dim, Kernel)
        self.norm = _RowLayerNorm(dim)

    def run(self, x, vset):
        return self.SConv3d.run_ln(x, vset, self.norm, relu=True, residual=x)

    def forward(self, x, coords, spitial_shape, bs):
        return self.run(x.contiguous(), voxel_set_for(coords))


class TrainBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d parameters; batch statistics over all rows, evaluated by the HIP kernels.
    The reference keeps every BatchNorm in train mode at test time (main.py:357); running
    statistics are therefore never read and are not tracked here."""

    def run(self, x, residual=None, relu=False, out=None):
        if recording():
            # like the reference in train mode the running statistics are updated (and never read)
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, True, self.momentum, self.eps)
            y = y + residual if residual is not None else y
            y = F.relu(y) if relu else y
            return y if out is None else out.copy_(y)
        return SP.batchnorm_train(x, self.weight, self.bias, self.eps, residual, relu, out)

    def run_partials(self, x, partial, residual=None, relu=False, out=None):
        """second half of the BatchNorm from the producing convolution's summaries"""
        return SP.batchnorm_apply_partials(x, partial, self.weight, self.bias, self.eps, residual, relu, out)

    def pending_affine(self, partial):
        """the BatchNorm in affine form (scale, shift) from the producer's summaries: consumers apply it on load"""
        return SP.bn_affine(partial, self.weight, self.bias, self.eps)


# ------------------------------------------------------------------------------------------------
# dense blocks (PyTorch-ROCm convolutions; on the GPU inference path the train-mode BatchNorm2d of
# channels-last activations runs on the HIP BatchNorm kernels: a [N,C,H,W] channels-last tensor IS a
# row-major [N*H*W, C] matrix, and MIOpen's spatial BN costs ~26 us per call on these small maps)
# ------------------------------------------------------------------------------------------------
def _rows(x):
    """channels-last [N,C,H,W] -> its [N*H*W, C] row-major view (no copy)"""
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c)


def _hip_bn_ok(x):
    return (x.is_cuda and not torch.is_grad_enabled() and x.dim() == 4 and x.shape[1] <= 256
            and x.is_contiguous(memory_format=torch.channels_last) and x.dtype == torch.float32)


def bn2d_train(bn, x, relu=False):
    """train-mode BatchNorm2d (+ReLU): HIP kernels for channels-last CUDA tensors under no_grad,
    else the PyTorch module"""
    if _hip_bn_ok(x):
        rows = _rows(x)
        SP.batchnorm_train(rows, bn.weight, bn.bias, bn.eps, relu=relu, out=rows)
        return x
    y = bn(x)
    return F.relu(y) if relu else y


def upsample2x_bilinear(x):
    """F.interpolate(x, scale_factor=2, mode="bilinear"); HIP kernel for channels-last CUDA tensors"""
    if x.is_cuda and not torch.is_grad_enabled() and x.shape[1] % 4 == 0 and x.dtype == torch.float32 \
            and x.is_contiguous(memory_format=torch.channels_last):
        from . import _lib
        n, c, h, w = x.shape
        out = torch.empty((n, c, 2 * h, 2 * w), dtype=x.dtype, device=x.device,
                          memory_format=torch.channels_last)
        _lib.check(_lib.load().eprecon_upsample2x_nhwc_async(_lib.ptr(x), _lib.ptr(out), n, h, w, c,
                                                             _lib.current_stream()), "eprecon_upsample2x_nhwc_async")
        return out
    return F.interpolate(x, scale_factor=2, mode="bilinear")
class Conv2d_Block(nn.Module):
    """conv(k, same) -> BN -> ReLU (models/modules.py:372-383)"""

    def __init__(self, C_in, C_out, Kernel):
        super().__init__()
        self.conv = nn.Conv2d(C_in, C_out, Kernel, padding="same")
        self.bn = nn.BatchNorm2d(C_out)
        self.act = nn.ReLU()

    def forward(self, x):
        return bn2d_train(self.bn, self.conv(x), relu=True)

    def run_act(self, x, grid, out=None, aff=None):
        """D2.Act -> D2.Act on the HIP gather-GEMM path (dense2d.py): one launch, BatchNorm left pending"""
        return D2.conv_bn_act(self.conv, self.bn, x, grid, out=out, aff=aff, relu=True)

    def run_rows(self, x, grid, out=None):
        """plain pixel rows [V*H*W, C_in] -> materialised [V*H*W, C_out]"""
        return D2.materialize(self.run_act(D2.Act(x), grid), out=out)


class Conv2d_Residual_Block(nn.Module):
    """BN(x + ReLU(conv(x))) (models/modules.py:385-399)"""

    def __init__(self, C, Kernel):
        super().__init__()
        self.conv = nn.Conv2d(C, C, Kernel, padding="same")
        self.bn = nn.BatchNorm2d(C)
        self.relu = nn.ReLU()

    def forward(self, x):
        return bn2d_train(self.bn, x + self.relu(self.conv(x)))

    def run_act(self, x, grid):
        return D2.conv_bn_act(self.conv, self.bn, x, grid, relu=False, pre_relu=True, residual=x)

    def run_rows(self, x, grid, out=None):
        return D2.materialize(self.run_act(D2.Act(x), grid), out=out)


class ELAN(nn.Module):
    """dense 2D ELAN (models/modules.py:340-370)"""

    def __init__(self, dim):
        super().__init__()
        h = dim // 2
        self.conv1 = Conv2d_Block(dim, dim, 1)
        self.conv2 = Conv2d_Block(dim, dim, 1)
        self.conv3 = Conv2d_Block(dim, h, 3)
        self.conv4 = Conv2d_Block(h, h, 3)
        self.conv5 = Conv2d_Block(h, h, 3)
        self.conv6 = Conv2d_Block(h, h, 3)
        self.conv7 = Conv2d_Block(dim * 4, dim, 1)

    def forward(self, x):
        parts = [self.conv1(x), self.conv2(x)]
        for layer in (self.conv3, self.conv4, self.conv5, self.conv6):
            parts.append(layer(parts[-1]))
        return self.conv7(torch.cat(parts, dim=1))

    def run_act(self, x, grid):
        """the six branches write raw outputs + their pending BatchNorms straight into one concat
        buffer and its (scale, shift) vectors; conv7 applies them while gathering"""
        d = x.rows.shape[1]
        h = d // 2
        dev = x.rows.device
        cat = torch.empty((x.rows.shape[0], 4 * d), dtype=torch.float32, device=dev)
        if D2._ARENA is not None:      # BatchNorm form (c): one accumulator block for the concat buffer, a slice per branch
            acc = D2.AccSlice.new(4 * d)
            eps = self.conv1.bn.eps
            part = lambda a, b: D2.Act(cat[:, a:b], relu=True, acc=acc.part(a, b), eps=eps)
            sl = lambda a, b: dict(out=cat[:, a:b], aff=acc.part(a, b))
        else:
            aff = torch.empty((2, 4 * d), dtype=torch.float32, device=dev)
            part = lambda a, b: D2.Act(cat[:, a:b], aff[0, a:b], aff[1, a:b], True)
            sl = lambda a, b: dict(out=cat[:, a:b], aff=(aff[0, a:b], aff[1, a:b]))
        self.conv1.run_act(x, grid, **sl(0, d))
        self.conv2.run_act(x, grid, **sl(d, 2 * d))
        self.conv3.run_act(part(d, 2 * d), grid, **sl(2 * d, 2 * d + h))
        self.conv4.run_act(part(2 * d, 2 * d + h), grid, **sl(2 * d + h, 3 * d))
        self.conv5.run_act(part(2 * d + h, 3 * d), grid, **sl(3 * d, 3 * d + h))
        self.conv6.run_act(part(3 * d, 3 * d + h), grid, **sl(3 * d + h, 4 * d))
        return self.conv7.run_act(part(0, 4 * d), grid)

    def run_rows(self, x, grid, out=None):
        return D2.materialize(self.run_act(D2.Act(x), grid), out=out)


class Fusion_Block(nn.Module):
    """3x3 conv-BN-ReLU, 1x1 conv-BN-ReLU, ELAN (models/modules.py:313-338)"""

    def __init__(self, C):
        super().__init__()
        self.conv1 = nn.Conv2d(C, C, 3, padding="same")
        self.bn1 = nn.BatchNorm2d(C)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(C, C, 1, padding="same")
        self.bn2 = nn.BatchNorm2d(C)
        self.ELAN = ELAN(C)

    def forward(self, x):
        x = bn2d_train(self.bn1, self.conv1(x), relu=True)
        x = bn2d_train(self.bn2, self.conv2(x), relu=True)
        return self.ELAN(x)

    def run_rows(self, x, grid, out=None):
        """plain rows in, materialised rows out; 9 convolution launches + 1 affine launch"""
        a = D2.conv_bn_act(self.conv1, self.bn1, D2.Act(x), grid, relu=True)
        a = D2.conv_bn_act(self.conv2, self.bn2, a, grid, relu=True)
        return D2.materialize(self.ELAN.run_act(a, grid), out=out)


def _ln_rows(norm, y, residual=None, pre_relu=False, post_relu=False):
    """[ReLU]( LayerNorm( [ReLU](y) [+ residual] ) ) for the heads: one HIP row-wise kernel on the GPU inference
    path (in place on the Linear output), the PyTorch ops otherwise"""
    if y.is_cuda and not torch.is_grad_enabled() and y.dim() == 2 and y.dtype == torch.float32 and y.is_contiguous() \
            and (residual is None or (residual.is_contiguous() and residual.dtype == torch.float32)):
        return SP.rowwise_layernorm(y, norm.weight, norm.bias, norm.eps, residual, pre_relu, post_relu, out=y)
    t = F.relu(y) if pre_relu else y
    if residual is not None:
        t = t + residual
    t = norm(t)
    return F.relu(t) if post_relu else t


# EPRECON_FUSED_HEADS=0: the heads as separate Linear / LayerNorm launches (the PyTorch modules) instead of csrc/heads.hip
_FUSED_HEADS = __import__("os").environ.get("EPRECON_FUSED_HEADS", "1") == "1"


def fused_heads_ok(mod, x):
    """inference on the GPU: the whole Linear4xTrans as one launch (eprecon_mlp4x_async)"""
    # (x.shape[1] == in_features: the kernel reads the first C columns of a wider row, where the PyTorch modules — and the
    # reference — raise on the shape mismatch; a wider input takes the module path and raises there)
    return (_FUSED_HEADS and not torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and x.stride(1) == 1 and x.shape[1] == mod.linear1.in_features
            and SP.mlp4x_supported(mod.linear1.in_features, mod.linear3.out_features))


def linear4x_pair(mod_a, mod_b, x):
    """(mod_a(x), mod_b(x)) for two heads of equal shape on the same rows (tsdf_preds / occ_preds,
    models/neucon_network.py:437-438): one launch on the GPU inference path"""
    if fused_heads_ok(mod_a, x) and mod_a.linear3.out_features == mod_b.linear3.out_features \
            and mod_a.linear1.in_features == mod_b.linear1.in_features:
        ya, yb = SP.mlp4x([mod_a, mod_b], x)
        return ya, yb
    return mod_a(x), mod_b(x)


class Linear4xTrans(nn.Module):
    """Linear(C,4C)-LN-ReLU, Linear(4C,C)-LN-ReLU, Linear(C,C_out) (+ skip when C == C_out)
    (models/modules.py:273-311); xavier-uniform weights, zero biases."""

    def __init__(self, C_in, C_out):
        super().__init__()
        self.linear1 = nn.Linear(C_in, C_in * 4)
        self.norm1 = nn.LayerNorm(C_in * 4)
        self.relu = nn.ReLU()
        self.linear2 = nn.Linear(C_in * 4, C_in)
        self.norm2 = nn.LayerNorm(C_in)
        self.linear3 = nn.Linear(C_in, C_out)
        self.use_residual = C_in == C_out
        for lin in (self.linear1, self.linear2, self.linear3):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, x):
        if recording() and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
            # training: the three per-voxel GEMMs and their weight gradients ([C, N] x [N, C'] with N > 100k rows, which
            # rocBLAS runs on 32 x 32 tiles) on the HIP gather-GEMM / wgrad kernels
            lin = lambda layer, t: AG.sparse_conv(t, layer.weight.t(), None, layer.bias)
        else:
            lin = lambda layer, t: layer(t)
        if fused_heads_ok(self, x):
            return SP.mlp4x([self], x)[0]
        h = _ln_rows(self.norm1, lin(self.linear1, x), post_relu=True)
        h = _ln_rows(self.norm2, lin(self.linear2, h), post_relu=True)
        y = lin(self.linear3, h)
        return y + h if self.use_residual else y


class Linear_Residual(nn.Module):
    """LN(x + ReLU(Linear(x))) (models/modules.py:454-467)"""

    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(dim, dim)
        self.activation = nn.ReLU()
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return _ln_rows(self.norm, self.linear(x), residual=x, pre_relu=True)


class Panoptic_Feat_Fusion(nn.Module):
    """Only `generate_mask_features` is on the reference's path (models/modules.py:574-580,
    called at models/neucon_network.py:557): three SparseConv3d_Residual(48) layers on the
    finest-level voxels.  The unused linear layers are kept so the state_dict keys match."""

    def __init__(self, self_channel, panoptic_channel, ch_initialization):
        super().__init__()
        self.img2panoptic_0 = nn.Linear(ch_initialization[2], panoptic_channel)
        self.occ2panoptic_0 = nn.Linear(self_channel, panoptic_channel)
        self.pre_fusion = nn.Linear(panoptic_channel * 2, panoptic_channel)
        self.pre_fusion_0 = Linear_Residual(panoptic_channel)
        self.pre_fusion_1 = Linear_Residual(panoptic_channel)
        self.mask_feat_extraction_0 = SparseConv3d_Residual(panoptic_channel, 3)
        self.mask_feat_extraction_1 = SparseConv3d_Residual(panoptic_channel, 3)
        self.mask_feat_extraction_2 = SparseConv3d_Residual(panoptic_channel, 3)

    def generate_mask_features(self, panoptic_feats, coords_b, coords_xyz, batch_size, spitial_shape):
        coords = torch.cat([coords_b.unsqueeze(1), coords_xyz], dim=1).to(torch.int32).contiguous()
        vset = SP.VoxelSet(coords, 1)
        x = panoptic_feats.contiguous()
        for layer in (self.mask_feat_extraction_0, self.mask_feat_extraction_1, self.mask_feat_extraction_2):
            x = layer.run(x, vset)
        return x


# ------------------------------------------------------------------------------------------------
# torchsparse-style layers: point-voxel U-Net (SPVCNN) and sparse ConvGRU
# ------------------------------------------------------------------------------------------------
from .tensor import PointTensor, SparseTensor  # noqa: E402
from .torchsparse_utils import devoxelize_gate, initial_voxelize, point_to_voxel, voxel_to_point  # noqa: E402

__all__ = ["SPVCNN", "SConv3d", "ConvGRU"]


def _linear_wt(lin):
    """weight of an nn.Linear as the [C_in, C_out] matrix the per-voxel GEMM takes, transposed once per weight
    version (the inference path calls every layer thousands of times with the same weights)"""
    hit = getattr(lin, "_wt_cache", None)
    tag = (lin.weight._version, lin.weight.data_ptr())
    if hit is None or hit[0] != tag or hit[1].device != lin.weight.device:
        hit = (tag, lin.weight.detach().t().contiguous())
        lin._wt_cache = hit
    return hit[1]


class Conv3d(nn.Module):
    """spnn.Conv3d(inc, outc, kernel_size, stride, transposed), bias-free.  Parameter `kernel` is
    [K^3, inc, outc] ([inc, outc] when K = 1) like torchsparse's, default init uniform(+-1/sqrt(fan))
    with fan = K^3 * (outc if transposed else inc)."""

    def __init__(self, inc, outc, kernel_size=3, stride=1, dilation=1, transposed=False):
        super().__init__()
        assert dilation == 1 and (kernel_size, stride) in ((1, 1), (3, 1), (2, 2))
        self.kernel_size, self.stride, self.transposed = kernel_size, stride, transposed
        kvol = kernel_size ** 3
        shape = (inc, outc) if kvol == 1 else (kvol, inc, outc)
        self.kernel = nn.Parameter(torch.empty(shape))
        std = 1.0 / math.sqrt((outc if transposed else inc) * kvol)
        with torch.no_grad():
            self.kernel.uniform_(-std, std)

    def run(self, feats, nbr, out=None):
        if recording():
            y = AG.sparse_conv(feats, self.kernel, nbr)
            return y if out is None else out.copy_(y)
        return SP.sparse_conv(feats, self.kernel, nbr, None, out=out)

    def run_stats(self, feats, nbr, out=None, in_affine=None):
        """conv + the BatchNorm summaries of its output in one launch -> (y, partial)"""
        return SP.conv_stats(feats, self.kernel, nbr, in_affine=in_affine, out=out)


class BasicConvolutionBlock(nn.Module):
    """conv -> BN -> ReLU (models/modules.py:15-28)"""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(Conv3d(inc, outc, ks, stride, dilation), TrainBatchNorm1d(outc), nn.ReLU(True))

    def run(self, feats, nbr, out=None):
        if recording():
            return self.net[1].run(self.net[0].run(feats, nbr), relu=True)
        y, partial = self.net[0].run_stats(feats, nbr, out=out)
        return self.net[1].run_partials(y, partial, relu=True, out=y)


class BasicDeconvolutionBlock(nn.Module):
    """transposed conv -> BN -> ReLU (models/modules.py:31-43)"""

    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = nn.Sequential(Conv3d(inc, outc, ks, stride, transposed=True), TrainBatchNorm1d(outc),
                                 nn.ReLU(True))

    def run(self, feats, nbr, out=None):
        if recording():
            return self.net[1].run(self.net[0].run(feats, nbr), relu=True)
        y, partial = self.net[0].run_stats(feats, nbr, out=out)
        return self.net[1].run_partials(y, partial, relu=True, out=y)


class ResidualBlock(nn.Module):
    """ReLU( [conv3-BN-ReLU-conv3-BN](x) + [identity | conv1-BN](x) )  (models/modules.py:46-72)"""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        assert stride == 1
        self.net = nn.Sequential(Conv3d(inc, outc, ks, stride, dilation), TrainBatchNorm1d(outc), nn.ReLU(True),
                                 Conv3d(outc, outc, ks, 1, dilation), TrainBatchNorm1d(outc))
        self.downsample = nn.Sequential() if inc == outc else nn.Sequential(
            Conv3d(inc, outc, 1, 1), TrainBatchNorm1d(outc))
        self.relu = nn.ReLU(True)

    def run(self, feats, nbr, out=None):
        if recording():
            y = self.net[1].run(self.net[0].run(feats, nbr), relu=True)
            y = self.net[3].run(y, nbr)
            skip = feats if len(self.downsample) == 0 else self.downsample[1].run(self.downsample[0].run(feats, None))
            return self.net[4].run(y, residual=skip, relu=True)
        # conv1's BatchNorm + ReLU stays pending and is applied by conv2 while it gathers
        y, p1 = self.net[0].run_stats(feats, nbr)
        scale, shift = self.net[1].pending_affine(p1)
        y2, p2 = self.net[3].run_stats(y, nbr, in_affine=(scale, shift, True))
        if len(self.downsample) == 0:
            return self.net[4].run_partials(y2, p2, residual=feats, relu=True, out=out if out is not None else y2)
        # the 1x1 skip convolution's BatchNorm stays pending as well: the block's tail applies both in one pass
        skip, ps = self.downsample[0].run_stats(feats, None)
        return SP.batchnorm_apply_partials(y2, p2, self.net[4].weight, self.net[4].bias, self.net[4].eps, residual=skip,
                                           relu=True, out=out if out is not None else y2,
                                           res_affine=self.downsample[1].pending_affine(ps))


class _PointMLP(nn.Sequential):
    """nn.Linear -> BatchNorm1d (train) -> ReLU on point features (models/modules.py:125-136)"""

    def __init__(self, inc, outc):
        super().__init__(nn.Linear(inc, outc), TrainBatchNorm1d(outc), nn.ReLU(True))

    def run(self, feats):
        # the Linear's bias cancels in the train-mode BatchNorm that follows (it shifts the batch mean by the same
        # amount), so the bias-free convolution with the statistics epilogue gives the same result in one pass less
        lin = self[0]
        if recording():   # (per-point GEMM on the HIP kernel: rocBLAS picks 32x32 tiles for these [N, <100] x [<100, <100] shapes)
            return self[1].run(AG.sparse_conv(feats, lin.weight.t(), None, lin.bias), relu=True)
        y, partial = SP.conv_stats(feats, _linear_wt(lin), None)
        return self[1].run_partials(y, partial, relu=True, out=y)


# EPRECON_SPVCNN_NATIVE=0: the body of an SPVCNN pass issued launch by launch from Python (the round-4 path) instead of by ONE
# library call (eprecon_spvcnn_forward_async, csrc/spvcnn_forward.hip): the same entry points, descriptors and order —
# bit-identical results; ~115 launches whose host cost drops from ~15 us to ~3 us each (the host runs further ahead; the
# GPU-bound fragment itself is not measurably shorter: DESIGN.md 7g)
_NATIVE_SPVCNN = __import__("os").environ.get("EPRECON_SPVCNN_NATIVE", "1") == "1"


class SPVCNN(nn.Module):
    """Point-voxel U-Net (models/modules.py:75-175): stem, two k2s2 down stages with residual blocks,
    two transposed up stages with skip concatenation, three voxel<->point transfers, two point MLPs.
    Every sparse op runs in libeprecon_hip.so; the three kernel maps (tensor strides 1, 2, 4) and the
    two strided maps are built once per forward and shared by all 25 convolutions."""

    def __init__(self, **kwargs):
        super().__init__()
        self.dropout = kwargs["dropout"]
        cr = kwargs.get("cr", 1.0)
        cs = [int(cr * c) for c in (32, 64, 128, 96, 96)]
        self.cs = cs
        self.pres, self.vres = kwargs["pres"], kwargs["vres"]
        self.stem = nn.Sequential(Conv3d(kwargs["in_channels"], cs[0], 3, 1), TrainBatchNorm1d(cs[0]), nn.ReLU(True))
        self.stage1 = nn.Sequential(BasicConvolutionBlock(cs[0], cs[0], ks=2, stride=2),
                                    ResidualBlock(cs[0], cs[1]), ResidualBlock(cs[1], cs[1]))
        self.stage2 = nn.Sequential(BasicConvolutionBlock(cs[1], cs[1], ks=2, stride=2),
                                    ResidualBlock(cs[1], cs[2]), ResidualBlock(cs[2], cs[2]))
        self.up1 = nn.ModuleList([BasicDeconvolutionBlock(cs[2], cs[3], ks=2, stride=2),
                                  nn.Sequential(ResidualBlock(cs[3] + cs[1], cs[3]), ResidualBlock(cs[3], cs[3]))])
        self.up2 = nn.ModuleList([BasicDeconvolutionBlock(cs[3], cs[4], ks=2, stride=2),
                                  nn.Sequential(ResidualBlock(cs[4] + cs[0], cs[4]), ResidualBlock(cs[4], cs[4]))])
        self.point_transforms = nn.ModuleList([_PointMLP(cs[0], cs[2]), _PointMLP(cs[2], cs[4])])
        assert not self.dropout, "SPARSEREG.DROPOUT is False in the reference configs (config/default.py:69)"

    def _forward_recording(self, z):
        """the same network through the recording operators (training): concatenations are torch.cat, every
        sparse op is an autograd Function over the HIP kernels"""
        x0 = initial_voxelize(z, self.pres, self.vres)
        s1 = x0.vset
        s2, down12, up21 = s1.downsample()
        s4, down24, up42 = s2.downsample()
        k1, k2, k4 = s1.kernel_map(3), s2.kernel_map(3), s4.kernel_map(3)
        f0 = self.stem[1].run(self.stem[0].run(x0.F, k1), relu=True)
        x0 = SparseTensor(f0, s1)
        z0 = voxel_to_point(x0, z)
        x1 = point_to_voxel(x0, z0)
        f1 = self.stage1[2].run(self.stage1[1].run(self.stage1[0].run(x1.F, down12), k2), k2)
        f2 = self.stage2[2].run(self.stage2[1].run(self.stage2[0].run(f1, down24), k4), k4)
        x2 = SparseTensor(f2, s4)
        z1 = voxel_to_point(x2, z0, out=self.point_transforms[0].run(z0.F), accumulate=True)
        y3 = point_to_voxel(x2, z1)
        f = torch.cat([self.up1[0].run(y3.F, up42), f1], dim=1)
        f = self.up1[1][1].run(self.up1[1][0].run(f, k2), k2)
        f = torch.cat([self.up2[0].run(f, up21), f0], dim=1)
        f = self.up2[1][1].run(self.up2[1][0].run(f, k1), k1)
        z3 = voxel_to_point(SparseTensor(f, s1), z1, out=self.point_transforms[1].run(z1.F), accumulate=True)
        return z3.F

    def _apply(self, fn, *args, **kwargs):
        self._native_params = self._native = None      # .to() / .cuda() may replace the parameter objects
        return super()._apply(fn, *args, **kwargs)

    def _native_slots(self):
        """(conv modules, BatchNorm modules) in the slot order of eprecon_spvcnn_forward_desc"""
        convs, bns = [], []

        def basic(block):
            convs.append(block.net[0].kernel); bns.append(block.net[1])

        def res(block):
            convs.extend([block.net[0].kernel, block.net[3].kernel]); bns.extend([block.net[1], block.net[4]])
            if len(block.downsample):
                convs.append(block.downsample[0].kernel); bns.append(block.downsample[1])

        convs.append(self.stem[0].kernel); bns.append(self.stem[1])
        for stage in (self.stage1, self.stage2):
            basic(stage[0]); res(stage[1]); res(stage[2])
            if stage is self.stage2:
                convs.append(_linear_wt(self.point_transforms[0][0])); bns.append(self.point_transforms[0][1])
        for up in (self.up1, self.up2):
            basic(up[0]); res(up[1][0]); res(up[1][1])
        convs.append(_linear_wt(self.point_transforms[1][0])); bns.append(self.point_transforms[1][1])
        return convs, bns

    def _native_desc(self, device):
        """the static part of eprecon_spvcnn_forward_desc (weights, their operand-order packings, BatchNorm parameters), built
        once per parameter version"""
        from . import _lib
        slots = getattr(self, "_native_params", None)
        if slots is None:       # (walking the module tree costs ~0.25 ms per call: once; what is kept is WHERE the parameters
            # live — each submodule's _parameters dict + name —, so a Parameter object replaced later (load_state_dict(assign=True),
            # `module.weight = nn.Parameter(..)`) is seen by the key below: ADVICE r05)
            slots = self._native_params = [(m._parameters, n) for m in self.modules() for n in m._parameters if m._parameters[n] is not None]
        key = (device, tuple((id(p_), p_._version, p_.data_ptr()) for p_ in (d_[n] for d_, n in slots)))
        hit = getattr(self, "_native", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        convs, bns = self._native_slots()
        if len(convs) != _lib.SPVCNN_CONVS:       # (a channel plan whose residual blocks lost / gained a 1x1 skip: Python path)
            self._native = (key, None)
            return None
        d, keep = _lib.SpvcnnForwardDesc(), []
        with torch.no_grad():
            for slot, (w, bn) in enumerate(zip(convs, bns)):
                w3 = w if w.dim() == 3 else w.unsqueeze(0)
                kvol, ci, co = w3.shape
                wc = w3.detach().contiguous()
                e = d.conv[slot]
                e.weight, e.kvol, e.cin, e.cout = wc.data_ptr(), kvol, ci, co
                keep.append(wc)
                if kvol == 27:
                    pw = SP.packed_weight(w)
                    e.packed_weight = pw.data_ptr()
                    keep.append(pw)
                if kvol in (27, 1) and co <= SP.DIRECT_MAX_COUT:
                    pw16 = SP.packed_weight16(w3, owner=w)
                    e.packed_weight16 = pw16.data_ptr()
                    keep.append(pw16)
                b = d.bn[slot]
                b.gamma, b.beta, b.eps = bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps)
        d.cs[:] = self.cs
        self._native = (key, (d, keep))
        return self._native[1]

    def _forward_native(self, z):
        """SPVCNN.forward as ONE library call on the voxelisation of z (found in / entered into the cache like initial_voxelize
        does); None when a piece it needs is missing (the Python-issued pass then runs)"""
        import ctypes
        from . import _lib
        from . import torchsparse_utils as TU
        feat = z.F
        if feat.dtype != torch.float32 or feat.stride(1) != 1:
            return None
        pack = self._native_desc(feat.device)
        if pack is None:
            return None
        d, _keep = pack
        pts = z.C if z.C.is_contiguous() else z.C.contiguous()
        res = float(self.vres) / float(self.pres) if self.pres != 1 else float(self.vres)
        e = TU._voxelize_points(pts, res, 3)
        s1 = e.vset
        if getattr(e, "stride4", None) is None or s1._down is None or s1._down[0]._down is None or s1._k3 is None:
            return None
        s2, down12, up21 = s1._down
        s4, down24, up42 = s2._down
        if s2._k3 is None or s4._k3 is None or e.idx8 is None:
            return None
        lib = _lib.load()
        n = feat.shape[0]
        d.n, d.n1, d.n2, d.n4 = n, s1.n, s2.n, s4.n
        d.cin, d.feat, d.ld_feat = feat.shape[1], feat.data_ptr(), feat.stride(0)
        idx4, (offsets4, order4), idx8_4, w8_4 = e.stride4
        tensors = {"offsets1": e.lists[0], "order1": e.lists[1], "offsets4": offsets4, "order4": order4, "k1": s1._k3, "k2": s2._k3,
                   "k4": s4._k3, "down12": down12, "up21": up21, "down24": down24, "up42": up42, "idx8_1": e.idx8,
                   "weight8_1": e.w8, "idx8_4": idx8_4, "weight8_4": w8_4}
        for name, t in tensors.items():
            setattr(d, name, t.data_ptr())
        out = torch.empty((n, self.cs[4]), dtype=torch.float32, device=feat.device)
        d.out, d.ld_out = out.data_ptr(), out.stride(0)
        d.workspace, d.workspace_bytes = None, 0
        need = int(lib.eprecon_spvcnn_forward_workspace_bytes(ctypes.byref(d)))
        if need == 0:
            return None
        arena = torch.empty(need, dtype=torch.uint8, device=feat.device)
        d.workspace, d.workspace_bytes = arena.data_ptr(), need
        _lib.check(lib.eprecon_spvcnn_forward_async(ctypes.byref(d), _lib.current_stream()), "eprecon_spvcnn_forward_async")
        z.C, z.vox = e.scaled, e.vox      # (like initial_voxelize: the PointTensor now carries the scaled coordinates)
        return out

    def forward(self, z):
        if recording():
            return self._forward_recording(z)
        if _NATIVE_SPVCNN and z.F.is_cuda:
            out = self._forward_native(z)
            if out is not None:
                return out
        cs = self.cs
        dev = z.F.device
        x0 = initial_voxelize(z, self.pres, self.vres, levels=3)   # strides 1, 2, 4 numbered together: one host read
        s1 = x0.vset
        s2, down12, up21 = s1.downsample()
        s4, down24, up42 = s2.downsample()
        # concat buffers: the skip branches are written in place (torchsparse.cat for free)
        cat0 = torch.empty((s1.n, cs[4] + cs[0]), dtype=torch.float32, device=dev)
        cat1 = torch.empty((s2.n, cs[3] + cs[1]), dtype=torch.float32, device=dev)

        f0, p0 = self.stem[0].run_stats(x0.F, s1.kernel_map(3), out=cat0[:, cs[4]:])
        self.stem[1].run_partials(f0, p0, relu=True, out=f0)
        x0 = SparseTensor(f0, s1)
        z0 = voxel_to_point(x0, z)

        x1 = point_to_voxel(x0, z0)
        f = self.stage1[0].run(x1.F, down12)
        f = self.stage1[1].run(f, s2.kernel_map(3))
        f1 = self.stage1[2].run(f, s2.kernel_map(3), out=cat1[:, cs[3]:])
        f = self.stage2[0].run(f1, down24)
        f = self.stage2[1].run(f, s4.kernel_map(3))
        f2 = self.stage2[2].run(f, s4.kernel_map(3))
        x2 = SparseTensor(f2, s4)

        z1 = voxel_to_point(x2, z0, out=self.point_transforms[0].run(z0.F), accumulate=True)
        y3 = point_to_voxel(x2, z1)
        self.up1[0].run(y3.F, up42, out=cat1[:, :cs[3]])
        f = self.up1[1][0].run(cat1, s2.kernel_map(3))
        f = self.up1[1][1].run(f, s2.kernel_map(3))
        self.up2[0].run(f, up21, out=cat0[:, :cs[4]])
        f = self.up2[1][0].run(cat0, s1.kernel_map(3))
        f = self.up2[1][1].run(f, s1.kernel_map(3))
        y4 = SparseTensor(f, s1)
        z3 = voxel_to_point(y4, z1, out=self.point_transforms[1].run(z1.F), accumulate=True)
        return z3.F


class SConv3d(nn.Module):
    """voxelise -> Conv3d(k3) -> devoxelise, plus a point-wise Linear skip (models/modules.py:178-197).
    Like the reference it voxelises its input in place (z.C is overwritten by initial_voxelize)."""

    def __init__(self, inc, outc, pres, vres, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = Conv3d(inc, outc, ks, stride, dilation)
        self.point_transforms = nn.Sequential(nn.Linear(inc, outc))
        self.pres, self.vres = pres, vres

    def forward(self, z):
        x = initial_voxelize(z, self.pres, self.vres)
        y = SparseTensor(self.net.run(x.F, x.vset.kernel_map(3)), x.vset)
        lin = self.point_transforms[0]
        if recording():
            skip = AG.sparse_conv(z.F, lin.weight.t(), None, lin.bias)
        else:
            skip = SP.sparse_conv(z.F, _linear_wt(lin), None, lin.bias)
        return voxel_to_point(y, z, out=skip, accumulate=True)

    def run_gate(self, z, mode, h=None, zgate=None, out=None, skip=None, tail=None):
        """forward(z).F followed by the ConvGRU gate arithmetic of `mode` (torchsparse_utils.devoxelize_gate).
        skip: the point-wise Linear of this layer when the caller already computed it (ConvGRU: one GEMM for two gates)"""
        x = initial_voxelize(z, self.pres, self.vres)
        y = SparseTensor(self.net.run(x.F, x.vset.kernel_map(3)), x.vset)
        if skip is None:
            lin = self.point_transforms[0]
            skip = SP.sparse_conv(z.F, _linear_wt(lin), None, lin.bias)
        return devoxelize_gate(y, z, skip, mode, h=h, zgate=zgate, out=out, tail=tail)


class ConvGRU(nn.Module):
    """Sparse convolutional GRU cell (models/modules.py:200-222):
    z = sigma(convz([h,x])), r = sigma(convr([h,x])), q = tanh(convq([r*h, x])), h' = (1-z) h + z q."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128, pres=1, vres=1):
        super().__init__()
        self.convz = SConv3d(hidden_dim + input_dim, hidden_dim, pres, vres, 3)
        self.convr = SConv3d(hidden_dim + input_dim, hidden_dim, pres, vres, 3)
        self.convq = SConv3d(hidden_dim + input_dim, hidden_dim, pres, vres, 3)

    def forward(self, h, x, out=None):
        """`hx` is voxelised twice — convz, then convr on the coordinates convz already divided by vres (the
        in-place `z.C = ...` of ops/torchsparse_utils.py:33 with models/modules.py:216-217) — and convr
        devoxelises with convz's cached corner indices (torchsparse_utils.initial_voxelize, LITERAL_CONVR).
        `out` (optional, beyond the reference's signature): where the new hidden state is written."""
        if h.F.is_cuda and not torch.is_grad_enabled():
            return self._forward_fused(h, x, out)
        hx = PointTensor(torch.cat([h.F, x.F], dim=1), h.C)
        z = torch.sigmoid(self.convz(hx).F)
        r = torch.sigmoid(self.convr(hx).F)
        x.F = torch.cat([r * h.F, x.F], dim=1)
        q = torch.tanh(self.convq(x).F)
        h.F = (1 - z) * h.F + z * q
        if out is not None:
            out.copy_(h.F)
        return h.F

    def _forward_fused(self, h, x, out=None):
        """the same cell with the gate arithmetic in the devoxelisation kernels: r * h lands directly in the
        [r*h, x] concat buffer, the GRU mix is the epilogue of convq"""
        c = h.F.shape[1]
        hf, xf = h.F, x.F
        if (hf.stride(0) == hf.shape[1] + xf.shape[1] == xf.stride(0) and hf.stride(1) == 1 == xf.stride(1)
                and xf.data_ptr() == hf.data_ptr() + 4 * c):
            # h and x already sit side by side in one [N, c_h + c_x] buffer (GRUFusion gathers them that way)
            hx_f = torch.as_strided(hf, (hf.shape[0], hf.stride(0)), (hf.stride(0), 1))
        else:
            hx_f = torch.cat([hf, xf], dim=1)
            hf = hx_f[:, :c]
        hx = PointTensor(hx_f, h.C)
        # the point-wise Linear skips of convz and convr read the same [h, x] rows: ONE GEMM [N, c_in] x [c_in, 2 c]
        w_zr, b_zr = self._merged_skip()
        skip_zr = SP.sparse_conv(hx_f, w_zr, None, b_zr)
        z = self.convz.run_gate(hx, 1, skip=skip_zr[:, :c])
        # [r * h | x]: r * h written by convr's gate kernel, which copies the x half alongside (no clone of [h, x])
        rhx = torch.empty_like(hx_f)
        self.convr.run_gate(hx, 2, h=hf, out=rhx[:, :c], skip=skip_zr[:, c:], tail=(hx_f[:, c:], rhx[:, c:]))
        x.F = rhx
        h.F = self.convq.run_gate(x, 3, h=hf, zgate=z, out=out)
        return h.F

    def _merged_skip(self):
        lz, lr = self.convz.point_transforms[0], self.convr.point_transforms[0]
        tag = tuple((t._version, t.data_ptr()) for t in (lz.weight, lr.weight, lz.bias, lr.bias))
        hit = getattr(self, "_wt_cache", None)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                hit = (tag, torch.cat([lz.weight.t(), lr.weight.t()], dim=1).contiguous(), torch.cat([lz.bias, lr.bias]).contiguous())
            self._wt_cache = hit
        return hit[1], hit[2]
