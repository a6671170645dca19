"""Stage-wise CPU restatement of NeuConNet.forward's coarse-to-fine loop
(models/neucon_network.py:348-511,516-548) — test infrastructure.  Built from the other oracle
modules; the dense heads are restated in numpy (Linear4xTrans, models/modules.py:273-311).
Each function takes the stage's INPUTS (so that a test can check stage by stage: a near-zero
occupancy logit may legitimately flip between fp32 implementations and would otherwise change
every later voxel set)."""
import numpy as np

from . import pointvoxel as PV
from . import sparse as OS
from . import spvcnn as ON

F32 = np.float32


def _ln(x, g, b, eps=1e-5):
    x = x.astype(np.float64)
    mu = x.mean(1, keepdims=True)
    var = ((x - mu) ** 2).mean(1, keepdims=True)
    return ((x - mu) / np.sqrt(var + eps) * g + b).astype(F32)


def linear4x(sd, p, x):
    """models/modules.py:273-311"""
    h = x @ sd[p + ".linear1.weight"].T + sd[p + ".linear1.bias"]
    h = np.maximum(_ln(h, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]), 0)
    h = h @ sd[p + ".linear2.weight"].T + sd[p + ".linear2.bias"]
    h = np.maximum(_ln(h, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]), 0)
    y = h @ sd[p + ".linear3.weight"].T + sd[p + ".linear3.bias"]
    if sd[p + ".linear3.weight"].shape[0] == sd[p + ".linear3.weight"].shape[1]:
        y = y + h
    return y.astype(F32)


def sub_dict(sd, prefix):
    n = len(prefix) + 1
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


def spvcnn_stage(sd, i, coords, feat_in, origin, w2ac, voxel_size=0.04):
    """:387-402 -> (r_coords f32[N,4], feat_out)"""
    r = PV.aligned_coords(coords, origin, voxel_size, w2ac)
    vres = voxel_size * 2 ** (2 - i)
    return r, ON.spvcnn_forward(sub_dict(sd, f"sp_convs.{i}"), feat_in, r, 1, vres)


def heads_stage(sd, i, feat, threshold=0.0):
    """:414-415,454 -> (tsdf, occ, occupancy)"""
    tsdf = linear4x(sd, f"tsdf_preds.{i}", feat)
    occ = linear4x(sd, f"occ_preds.{i}", feat)
    return tsdf, occ, occ[:, 0] > threshold


def prune_to_ancestors(c0, c1, c2):
    """:516-542: level-1 / level-0 voxels that are ancestors of some level-2 voxel (per batch via
    the batch column of the key)"""
    d1 = OS.quantise(c2, 2)
    keep1 = OS.Index(d1).lookup(np.asarray(c1, np.int64)) >= 0
    d0 = OS.quantise(c2, 4)
    keep0 = OS.Index(d0).lookup(np.asarray(c0, np.int64)) >= 0
    return keep1, keep0
