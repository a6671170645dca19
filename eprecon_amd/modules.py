"""Building blocks of the 3D path — host-side mirror of models/modules.py of the reference.

Two families:
  * sparse layers, executed by libeprecon_hip.so through eprecon_amd.sparse:
      SparseSubMConv3d, SubMconv3dBlock, Spares3dELAN, SparseConv3d_Residual     (spconv in the reference)
      SPVCNN, SConv3d, ConvGRU and their blocks                                  (torchsparse in the reference;
                                                                                  see eprecon_amd/spvcnn.py)
  * small dense blocks that stay PyTorch-ROCm (MIOpen / rocBLAS), as BASELINE.json's north_star
    prescribes for the 2D side and the heads: Conv2d_Block, Conv2d_Residual_Block, ELAN,
    Fusion_Block, Linear4xTrans, Linear_Residual.

Class names, constructor arguments, forward signatures and parameter names follow the reference so
that its state_dict keys line up, except the sparse conv weights, whose layout is this build's
[K^3, C_in, C_out] with x-fastest offset order (spconv stores [C_out, kz, ky, kx, C_in];
`SparseSubMConv3d.load_spconv_weight` converts).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import sparse as SP

# ------------------------------------------------------------------------------------------------
# coordinate-set cache: the reference hands raw `coords` tensors to every layer; spconv rebuilds
# its indice pairs each time (indice_key=None).  Here the hash grid + 27-neighbour table is built
# once per distinct coords tensor and shared by all layers that see the same tensor.
# ------------------------------------------------------------------------------------------------
_SET_CACHE = []
_SET_CACHE_MAX = 8


def voxel_set_for(coords, stride=1):
    key = (coords.data_ptr(), coords.shape[0], coords._version, int(stride), coords.device)
    for k, ref, vs in _SET_CACHE:
        if k == key and ref is coords:
            return vs
    c = coords if coords.dtype == torch.int32 else coords.to(torch.int32)
    vs = SP.VoxelSet(c.contiguous(), stride)
    _SET_CACHE.append((key, coords, vs))
    if len(_SET_CACHE) > _SET_CACHE_MAX:
        _SET_CACHE.pop(0)
    return vs


def clear_voxel_set_cache():
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

    def load_spconv_weight(self, w):
        """w: spconv layout [C_out, kz, ky, kx, C_in] -> [K^3 (x fastest), C_in, C_out]"""
        k = self.kernel
        with torch.no_grad():
            self.weight.copy_(w.permute(1, 2, 3, 4, 0).reshape(k ** 3, w.shape[4], w.shape[0]))

    def run(self, features, vset, out=None, relu=False):
        nbr = vset.kernel_map(3) if self.kernel == 3 else None
        return SP.sparse_conv(features, self.weight, nbr, self.bias, out=out, relu=relu)

    def forward(self, features, coords, spitial_shape, bs):
        """features f32[N, C_in]; coords int[N, 4] (b,x,y,z) -> f32[N, C_out]"""
        return self.run(features.contiguous(), voxel_set_for(coords))


class _RowLayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters, evaluated by the fused row-wise HIP epilogue"""

    def run(self, x, residual=None, pre_relu=False, post_relu=False, out=None):
        return SP.rowwise_layernorm(x, self.weight, self.bias, self.eps, residual, pre_relu, post_relu, out)


class SubMconv3dBlock(nn.Module):
    """conv -> LayerNorm -> ReLU (models/modules.py:440-452)"""

    def __init__(self, C_in, C_out, Kernel, Stride, Padding):
        super().__init__()
        self.conv = SparseSubMConv3d(C_in, C_out, Kernel, Stride)
        self.ln = _RowLayerNorm(C_out)

    def run(self, x, vset, out=None):
        y = self.conv.run(x, vset, out=out)
        return self.ln.run(y, post_relu=True, out=y)


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
        y = self.SConv3d.run(x, vset)
        return self.norm.run(y, residual=x, pre_relu=True, out=y)

    def forward(self, x, coords, spitial_shape, bs):
        return self.run(x.contiguous(), voxel_set_for(coords))


class TrainBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d parameters; batch statistics over all rows, evaluated by the HIP kernels.
    The reference keeps every BatchNorm in train mode at test time (main.py:357); running
    statistics are therefore never read and are not tracked here."""

    def run(self, x, residual=None, relu=False, out=None):
        return SP.batchnorm_train(x, self.weight, self.bias, self.eps, residual, relu, out)


# ------------------------------------------------------------------------------------------------
# dense blocks (PyTorch-ROCm)
# ------------------------------------------------------------------------------------------------
class Conv2d_Block(nn.Module):
    """conv(k, same) -> BN -> ReLU (models/modules.py:372-383)"""

    def __init__(self, C_in, C_out, Kernel):
        super().__init__()
        self.conv = nn.Conv2d(C_in, C_out, Kernel, padding="same")
        self.bn = nn.BatchNorm2d(C_out)
        self.act = nn.ReLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Conv2d_Residual_Block(nn.Module):
    """BN(x + ReLU(conv(x))) (models/modules.py:385-399)"""

    def __init__(self, C, Kernel):
        super().__init__()
        self.conv = nn.Conv2d(C, C, Kernel, padding="same")
        self.bn = nn.BatchNorm2d(C)
        self.relu = nn.ReLU()

    def forward(self, x):
        return self.bn(x + self.relu(self.conv(x)))


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
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.relu(self.bn2(self.conv2(x)))
        return self.ELAN(x)


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
        h = self.relu(self.norm1(self.linear1(x)))
        h = self.relu(self.norm2(self.linear2(h)))
        y = self.linear3(h)
        return y + h if self.use_residual else y


class Linear_Residual(nn.Module):
    """LN(x + ReLU(Linear(x))) (models/modules.py:454-467)"""

    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(dim, dim)
        self.activation = nn.ReLU()
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(x + self.activation(self.linear(x)))


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
