"""Point <-> voxel glue of the point-voxel U-Net — mirror of ops/torchsparse_utils.py:15-105 on
libeprecon_hip.so (csrc/voxelize.hip, csrc/kernel_map.hip)."""
import torch

from . import _lib
from . import sparse as SP
from .tensor import PointTensor, SparseTensor

__all__ = ["initial_voxelize", "point_to_voxel", "voxel_to_point", "aligned_camera_coords"]


def aligned_camera_coords(coords, origin, voxel_size, world_to_aligned_camera):
    """models/neucon_network.py:387-398: int32[N,4] (b,x,y,z) voxel coords -> f32[N,4] (x,y,z,b)
    metric coordinates in the gravity-aligned middle-camera frame."""
    lib = _lib.load()
    coords = coords.contiguous() if coords.dtype == torch.int32 else coords.to(torch.int32).contiguous()
    origin = origin.float().reshape(-1, 3).contiguous()
    w2ac = world_to_aligned_camera.float().reshape(-1, 4, 4).contiguous()
    out = torch.empty((coords.shape[0], 4), dtype=torch.float32, device=coords.device)
    _lib.check(lib.eprecon_aligned_coords_async(_lib.ptr(coords), coords.shape[0], _lib.ptr(origin),
                                                origin.shape[0], float(voxel_size), _lib.ptr(w2ac),
                                                _lib.ptr(out), _lib.current_stream()),
               "eprecon_aligned_coords_async")
    return out


def _segment_lists(idx, m):
    lib = _lib.load()
    n, dev = idx.shape[0], idx.device
    offsets = torch.empty(m + 1, dtype=torch.int32, device=dev)
    order = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ws = _lib.workspace(lib.eprecon_segment_workspace_bytes(n, m), dev)
    _lib.check(lib.eprecon_segment_lists_async(_lib.ptr(idx), n, m, _lib.ptr(offsets), _lib.ptr(order),
                                               _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "eprecon_segment_lists_async")
    return offsets, order


def _segment_mean(feat, lists, m, out=None):
    lib = _lib.load()
    offsets, order = lists
    c = feat.shape[1]
    if out is None:
        # row pitch rounded up to 4 floats: the convolution that consumes these rows can then gather with
        # 16-byte loads even for the ragged channel counts of the SPVCNN stems (81 / 139 / 75 / 51)
        cp = (c + 3) & ~3
        out = torch.empty((m, cp), dtype=torch.float32, device=feat.device)[:, :c]
    _lib.check(lib.eprecon_segment_mean_async(_lib.ptr(feat), feat.stride(0), _lib.ptr(offsets),
                                              _lib.ptr(order), m, c, _lib.ptr(out), out.stride(0),
                                              _lib.current_stream()), "eprecon_segment_mean_async")
    return out


def initial_voxelize(z, init_res, after_res):
    """ops/torchsparse_utils.py:15-35: floor(z.C * init_res / after_res) -> unique voxels
    (first-occurrence order) -> scatter-mean of z.F.  Overwrites z.C with the scaled coordinates."""
    lib = _lib.load()
    pts = z.C.contiguous()
    n = pts.shape[0]
    res = float(after_res) / float(init_res) if init_res != 1 else float(after_res)
    scaled = torch.empty_like(pts)
    vox = torch.empty((n, 4), dtype=torch.int32, device=pts.device)
    _lib.check(lib.eprecon_point_quantize_async(_lib.ptr(pts), n, res, _lib.ptr(scaled), _lib.ptr(vox),
                                                _lib.current_stream()), "eprecon_point_quantize_async")
    uniq, inverse, grid = SP.unique_coords(vox, 1)
    vset = SP.VoxelSet(uniq, 1, grid=grid)
    lists = _segment_lists(inverse, vset.n)
    feat = _segment_mean(z.F, lists, vset.n)
    z.C, z.vox = scaled, vox
    # a new voxel set invalidates every per-stride lookup cached on the points.  (The reference
    # keeps them: a second SConv3d on the same PointTensor — ConvGRU's convr — devoxelises with the
    # FIRST voxelisation's indices into the SECOND voxel set, whose order is torchsparse's
    # hash order.  That cannot be reproduced and is not; see DESIGN.md "Known deviations".)
    z.idx_query.clear()
    z.weights.clear()
    z.additional_features["idx_query"].clear()
    z.additional_features["lists"].clear()
    z.additional_features["idx_query"][1] = inverse
    z.additional_features["lists"][1] = lists
    return SparseTensor(feat, vset)


def point_to_voxel(x, z, out=None):
    """ops/torchsparse_utils.py:40-63: scatter-mean of z.F into the voxels of x (tensor stride x.s)"""
    s = x.s
    lists = z.additional_features["lists"].get(s)
    if lists is None:
        idx = x.vset.grid.query(z.vox, quantum=s)
        lists = _segment_lists(idx, x.vset.n)
        z.additional_features["idx_query"][s] = idx
        z.additional_features["lists"][s] = lists
    return SparseTensor(_segment_mean(z.F, lists, x.vset.n, out), x.vset)


def voxel_to_point(x, z, nearest=False, out=None, accumulate=False):
    """ops/torchsparse_utils.py:68-105: trilinear interpolation of the voxel features of x at the
    points of z (weights renormalised over the corners that exist)."""
    assert not nearest  # never used with True by the reference
    lib = _lib.load()
    s = x.s
    n = z.C.shape[0]
    if s not in z.idx_query:
        idx8 = torch.empty((n, 8), dtype=torch.int32, device=z.C.device)
        w8 = torch.empty((n, 8), dtype=torch.float32, device=z.C.device)
        grid = x.vset.grid
        _lib.check(lib.eprecon_trilinear_map_async(_lib.ptr(grid.mem), grid.capacity, _lib.ptr(z.C), n, s,
                                                   _lib.ptr(idx8), _lib.ptr(w8), _lib.current_stream()),
                   "eprecon_trilinear_map_async")
        z.idx_query[s], z.weights[s] = idx8, w8
    c = x.F.shape[1]
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.F.device)
    _lib.check(lib.eprecon_devoxelize_async(_lib.ptr(x.F), x.F.stride(0), _lib.ptr(z.idx_query[s]),
                                            _lib.ptr(z.weights[s]), n, c, _lib.ptr(out), out.stride(0),
                                            int(accumulate), _lib.current_stream()), "eprecon_devoxelize_async")
    new = PointTensor(out, z.C, idx_query=z.idx_query, weights=z.weights)
    new.vox = z.vox
    new.additional_features = z.additional_features
    return new
