"""numpy restatement of the point <-> voxel transfers torchsparse provides to the reference
(ops/torchsparse_utils.py:15-105) — test infrastructure.  PARITY UNPINNED (torchsparse is not
vendored, see oracle/sparse.py); semantics per SURVEY.md appendix A.2, voxel numbering in
first-occurrence order, scatter-mean summed in point order."""
import ctypes

import numpy as np

from . import lib
from . import sparse as OS

F32 = np.float32


def aligned_coords(coords, origin, voxel_size, w2ac):
    """models/neucon_network.py:387-398 -> f32[N,4] (x,y,z,b)"""
    coords = np.ascontiguousarray(coords, np.int32)
    origin = np.ascontiguousarray(origin, F32).reshape(-1, 3)
    w2ac = np.ascontiguousarray(w2ac, F32).reshape(-1, 4, 4)
    out = np.zeros((coords.shape[0], 4), F32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib().eprecon_oracle_aligned_coords(p(coords), ctypes.c_int64(coords.shape[0]), p(origin),
                                        ctypes.c_int(origin.shape[0]), ctypes.c_float(voxel_size), p(w2ac), p(out))
    return out


def point_quantize(pts, res):
    """-> scaled f32[N,4] (x/res,y/res,z/res,b), voxel int32[N,4] (b, floor...)"""
    pts = np.asarray(pts, F32)
    scaled = pts.copy()
    scaled[:, :3] = pts[:, :3] / F32(res)
    fl = np.floor(scaled[:, :3]).astype(np.int32)
    vox = np.concatenate([pts[:, 3:4].astype(np.int32), fl], 1)
    return scaled, vox


def segment_mean(feat, idx, m):
    """scatter-mean in point order; rows with idx -1 are dropped; empty voxels -> 0"""
    feat = np.asarray(feat, F32)
    out = np.zeros((m, feat.shape[1]), F32)
    cnt = np.zeros(m, np.int64)
    live = idx >= 0
    np.add.at(out, idx[live], feat[live])
    np.add.at(cnt, idx[live], 1)
    nz = cnt > 0
    out[nz] = out[nz] / cnt[nz, None].astype(F32)
    return out


def trilinear(vox_coords, stride, scaled_pts):
    """idx int32[N,8], w f32[N,8] (k = 4bx + 2by + bz), SURVEY.md appendix A.2"""
    index = OS.Index(vox_coords)
    p = np.asarray(scaled_pts, F32)
    s = F32(stride)
    pf = np.floor(p[:, :3] / s) * s
    pc = pf + s
    base = pf.astype(np.int64)
    b = p[:, 3].astype(np.int64)
    n = len(p)
    idx = np.full((n, 8), -1, np.int32)
    w = np.zeros((n, 8), F32)
    for k in range(8):
        o = np.array([(k >> 2) & 1, (k >> 1) & 1, k & 1])
        q = np.concatenate([b[:, None], base + o[None] * stride], 1)
        idx[:, k] = index.lookup(q)
        f = [(p[:, a] - pf[:, a]) if o[a] else (pc[:, a] - p[:, a]) for a in range(3)]
        wk = (f[0] * f[1] * f[2]).astype(F32)
        if stride != 1:
            wk = wk / (s * s * s)
        w[:, k] = np.where(idx[:, k] >= 0, wk, F32(0))
    den = w.sum(1, dtype=F32) + F32(1e-8)
    return idx, (w / den[:, None]).astype(F32)


def devoxelize(vfeat, idx, w):
    vfeat = np.asarray(vfeat, F32)
    out = np.zeros((idx.shape[0], vfeat.shape[1]), F32)
    for k in range(8):
        live = idx[:, k] >= 0
        out[live] += w[live, k, None] * vfeat[idx[live, k]]
    return out


def sphash(coords_bxyz):
    """torchsparse F.sphash (ops/torchsparse_utils.py:19; kernel recalled from torchsparse 1.4,
    SURVEY.md appendix A.2): 64-bit FNV-1a over the int32 (x, y, z, batch) of a row, folded to 60 bits"""
    c = np.asarray(coords_bxyz, np.int32)[:, [1, 2, 3, 0]]
    h = np.full(len(c), 14695981039346656037, np.uint64)
    with np.errstate(over="ignore"):
        for j in range(4):
            h ^= c[:, j].astype(np.uint32).astype(np.uint64)
            h *= np.uint64(1099511628211)
    h = (h >> np.uint64(60)) ^ (h & np.uint64(0x0FFFFFFFFFFFFFFF))
    return h.astype(np.int64)


class Points:
    """PointTensor stand-in: F f32[N,C], C f32[N,4] (xyzb) + the voxel-unit coordinates cached by
    initial_voxelize (which overwrites z.C in the reference, ops/torchsparse_utils.py:33) and the
    per-stride idx_query / weights caches of voxel_to_point (ops/torchsparse_utils.py:69-71,94-96)"""

    def __init__(self, F, C):
        self.F, self.C = np.asarray(F, F32), np.asarray(C, F32)
        self.vox = None  # int32[N,4] bxyz at stride 1 (floor of the scaled coords)
        self.idx_query, self.weights = {}, {}


def initial_voxelize(z, init_res, after_res, hash_order=False):
    """-> (voxel coords int32[M,4], voxel feats f32[M,C], idx_query int32[N]); mutates z.C.
    hash_order: number the voxels by ascending sphash like the reference (`torch.unique(pc_hash)`,
    ops/torchsparse_utils.py:20) instead of by first occurrence.  The caches on z are NOT cleared (the
    reference never clears them), which is what makes a second SConv3d on the same points reuse the first
    one's corner indices."""
    res = F32(after_res) / F32(init_res) if init_res != 1 else F32(after_res)
    scaled, vox = point_quantize(z.C, res)
    uniq, inv = OS.unique_first(vox, 1)
    if hash_order:
        perm = np.argsort(sphash(uniq), kind="stable")
        rank = np.empty_like(perm)
        rank[perm] = np.arange(len(perm))
        uniq, inv = uniq[perm], rank[inv]
    z.C, z.vox = scaled, vox
    return uniq, segment_mean(z.F, inv, len(uniq)), inv


def point_to_voxel(vox_coords, stride, z, feat):
    q = OS.quantise(z.vox, stride)
    idx = OS.Index(vox_coords).lookup(q)
    return segment_mean(feat, idx, len(vox_coords))


def voxel_to_point(vox_coords, stride, vfeat, z, reuse_cached=False):
    """reuse_cached: behave like the reference's voxel_to_point, which takes z.idx_query[stride] /
    z.weights[stride] when present (ops/torchsparse_utils.py:69-71,97-99) whatever voxel set they were
    computed for; indices beyond the current set (out-of-bounds reads in the reference) contribute 0."""
    if reuse_cached and stride in z.idx_query:
        idx, w = z.idx_query[stride], z.weights[stride]
        idx = np.where(idx < len(vfeat), idx, -1)
    else:
        idx, w = trilinear(vox_coords, stride, z.C)
        if reuse_cached:
            z.idx_query[stride], z.weights[stride] = idx, w
    return devoxelize(vfeat, idx, w)
