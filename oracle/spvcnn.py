"""numpy restatement of SPVCNN / SConv3d / ConvGRU as wired by the reference
(models/modules.py:15-222) — test infrastructure, PARITY UNPINNED
for the torchsparse layer semantics (oracle/sparse.py)."""
import numpy as np

from . import pointvoxel as PV
from . import sparse as OS

F32 = np.float32


class Level:
    """one tensor stride of a sparse U-Net pass: coords + k3 map (+ maps to the next level)"""

    def __init__(self, coords, stride):
        self.coords, self.stride = coords, stride
        self.k3 = OS.kernel_map(coords, coords, 3, stride)

    def down(self):
        coarse, parent = OS.unique_first(self.coords, 2 * self.stride)
        nxt = Level(coarse, 2 * self.stride)
        return nxt, OS.kernel_map(self.coords, coarse, 2, self.stride), OS.transpose_map(self.coords, parent, self.stride)


def conv_bn(sd, p, x, nbr, relu):
    """spnn.Conv3d (no bias) + spnn.BatchNorm (train) [+ ReLU]: `p`.0 conv kernel, `p`.1 bn"""
    y = OS.sparse_conv(x, nbr, sd[p + ".0.kernel"])
    return OS.batchnorm_train(y, sd[p + ".1.weight"], sd[p + ".1.bias"], relu=relu)


def residual_block(sd, p, x, nbr):
    """models/modules.py:46-72"""
    y = OS.sparse_conv(x, nbr, sd[p + ".net.0.kernel"])
    y = OS.batchnorm_train(y, sd[p + ".net.1.weight"], sd[p + ".net.1.bias"], relu=True)
    y = OS.sparse_conv(y, nbr, sd[p + ".net.3.kernel"])
    if p + ".downsample.0.kernel" in sd:
        s = OS.sparse_conv(x, None, sd[p + ".downsample.0.kernel"])
        s = OS.batchnorm_train(s, sd[p + ".downsample.1.weight"], sd[p + ".downsample.1.bias"])
    else:
        s = x
    return OS.batchnorm_train(y, sd[p + ".net.4.weight"], sd[p + ".net.4.bias"], residual=s, relu=True)


def point_mlp(sd, p, f):
    """nn.Linear + BatchNorm1d (train) + ReLU (models/modules.py:125-136)"""
    y = f @ sd[p + ".0.weight"].T + sd[p + ".0.bias"]
    return OS.batchnorm_train(y, sd[p + ".1.weight"], sd[p + ".1.bias"], relu=True)


def spvcnn_forward(sd, feat, coords_xyzb, pres, vres):
    """models/modules.py:148-175; feat f32[N,Cin], coords f32[N,4] (metres, xyzb) -> f32[N, cs4]"""
    z = PV.Points(feat, coords_xyzb)
    c0, x0, _ = PV.initial_voxelize(z, pres, vres)
    l1 = Level(c0, 1)
    x0 = conv_bn(sd, "stem", x0, l1.k3, True)
    z0 = PV.voxel_to_point(l1.coords, 1, x0, z)
    x1 = PV.point_to_voxel(l1.coords, 1, z, z0)
    l2, down12, up21 = l1.down()
    x1 = conv_bn(sd, "stage1.0.net", x1, down12, True)
    x1 = residual_block(sd, "stage1.1", x1, l2.k3)
    x1 = residual_block(sd, "stage1.2", x1, l2.k3)
    l4, down24, up42 = l2.down()
    x2 = conv_bn(sd, "stage2.0.net", x1, down24, True)
    x2 = residual_block(sd, "stage2.1", x2, l4.k3)
    x2 = residual_block(sd, "stage2.2", x2, l4.k3)
    z1 = PV.voxel_to_point(l4.coords, 4, x2, z) + point_mlp(sd, "point_transforms.0", z0)
    y3 = PV.point_to_voxel(l4.coords, 4, z, z1)
    y3 = conv_bn(sd, "up1.0.net", y3, up42, True)
    y3 = np.concatenate([y3, x1], 1)
    y3 = residual_block(sd, "up1.1.0", y3, l2.k3)
    y3 = residual_block(sd, "up1.1.1", y3, l2.k3)
    y4 = conv_bn(sd, "up2.0.net", y3, up21, True)
    y4 = np.concatenate([y4, x0], 1)
    y4 = residual_block(sd, "up2.1.0", y4, l1.k3)
    y4 = residual_block(sd, "up2.1.1", y4, l1.k3)
    return PV.voxel_to_point(l1.coords, 1, y4, z) + point_mlp(sd, "point_transforms.1", z1)


def sconv3d(sd, p, z, pres, vres, literal=True):
    """models/modules.py:178-197: voxelise (overwrites z.C!) -> Conv3d k3 -> devoxelise + Linear(z.F).
    literal: voxels numbered in torchsparse's hash order and voxel_to_point reusing whatever corner
    indices are cached on z — what the reference does; False: every call uses its own indices."""
    c0, x, _ = PV.initial_voxelize(z, pres, vres, hash_order=literal)
    lvl = Level(c0, 1)
    x = OS.sparse_conv(x, lvl.k3, sd[p + ".net.kernel"])
    out = PV.voxel_to_point(lvl.coords, 1, x, z, reuse_cached=literal)
    return out + z.F @ sd[p + ".point_transforms.0.weight"].T + sd[p + ".point_transforms.0.bias"]


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def convgru(sd, p, h, x, coords_xyzb, pres, vres, literal=True):
    """models/modules.py:200-222.  `hx` is voxelised twice (convz, then convr on the coordinates
    convz already divided by vres) exactly as the reference's in-place z.C update implies; with
    `literal`, convr also devoxelises with the corner indices / weights convz cached on hx
    (ops/torchsparse_utils.py:70-71,97-99), which then address the SECOND voxel set in hash order."""
    hx = PV.Points(np.concatenate([h, x], 1), coords_xyzb)
    zg = _sigmoid(sconv3d(sd, p + ".convz", hx, pres, vres, literal))
    rg = _sigmoid(sconv3d(sd, p + ".convr", hx, pres, vres, literal))  # hx.C already scaled once
    xq = PV.Points(np.concatenate([rg * h, x], 1), coords_xyzb)
    q = np.tanh(sconv3d(sd, p + ".convq", xq, pres, vres, literal).astype(np.float64)).astype(F32)
    return ((1 - zg) * h + zg * q).astype(F32)
