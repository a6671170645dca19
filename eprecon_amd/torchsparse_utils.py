"""Point <-> voxel glue of the point-voxel U-Net — mirror of ops/torchsparse_utils.py:15-105 on
libeprecon_hip.so (csrc/voxelize.hip, csrc/kernel_map.hip)."""
import ctypes
import os

import torch

from . import _lib
from . import autograd as AG
from . import sparse as SP
from .tensor import PointTensor, SparseTensor

__all__ = ["initial_voxelize", "point_to_voxel", "voxel_to_point", "aligned_camera_coords", "devoxelize_gate",
           "clear_voxelization_cache", "SpvcnnPrefetch"]


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


def _segment_mean(feat, lists, m, out=None, idx=None):
    if torch.is_grad_enabled() and feat.requires_grad:
        # training: the recording form (eprecon_amd/autograd.py); `out` is a fusion of the inference path only
        assert idx is not None
        res = AG.segment_mean(feat, idx, lists, m)
        return res if out is None else out.copy_(res)
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


# ConvGRU's second gate convolution as the reference literally computes it (stale idx_query / weights of the first
# voxelisation applied to the second voxel set in torchsparse's hash order).  EPRECON_CONVGRU_LITERAL=0: every
# SConv3d devoxelises with the indices of its own voxelisation.
LITERAL_CONVR = os.environ.get("EPRECON_CONVGRU_LITERAL", "1") == "1"


class _VoxEntry:
    """One voxelisation of a point set at one resolution: everything that depends on the coordinates only
    (scaled points, voxel set + hash grid + kernel map, point -> voxel lists, trilinear corners / weights).
    The six SConv3d of the two ConvGRUs of a scale voxelise the same points: convz and convq of both GRUs
    share one entry, both convr the entry built on the already-scaled coordinates."""
    __slots__ = ("key", "pts", "scaled", "vox", "vset", "inverse", "lists", "idx8", "w8", "_order", "_stale", "stride4")

    def sphash_order(self):
        """(perm, rank): perm[k] = id of the voxel with the k-th smallest torchsparse hash, rank = its inverse —
        the order in which the reference numbers the voxels of initial_voxelize (ops/torchsparse_utils.py:19-21)"""
        if self._order is None:
            lib = _lib.load()
            c = self.vset.coords
            n = c.shape[0]
            perm = torch.empty(n, dtype=torch.int32, device=c.device)
            rank = torch.empty(n, dtype=torch.int32, device=c.device)
            ws = _lib.workspace(lib.eprecon_sphash_order_workspace_bytes(n), c.device)
            _lib.check(lib.eprecon_sphash_order_async(_lib.ptr(c), n, _lib.ptr(perm), _lib.ptr(rank), _lib.ptr(ws), ws.numel(),
                                                      _lib.current_stream()), "eprecon_sphash_order_async")
            self._order = (perm, rank)
        return self._order

    def stale_from(self, prev):
        """corner indices cached on `prev` (the first voxelisation of the same PointTensor) as the reference applies
        them to THIS voxel set: row k of the hash-ordered old set -> row k of the hash-ordered new set"""
        hit = self._stale.get(id(prev))
        if hit is None:
            lib = _lib.load()
            assert prev.idx8 is not None, "the first SConv3d has not devoxelised yet"
            perm_new, _ = self.sphash_order()
            _, rank_old = prev.sphash_order()
            out = torch.empty_like(prev.idx8)
            _lib.check(lib.eprecon_remap_index_async(_lib.ptr(prev.idx8), prev.idx8.numel(), _lib.ptr(rank_old),
                                                     _lib.ptr(perm_new), perm_new.shape[0], _lib.ptr(out),
                                                     _lib.current_stream()), "eprecon_remap_index_async")
            hit = (out, prev.w8, prev)   # `prev` kept alive: id() stays unique
            self._stale[id(prev)] = hit
        return hit[0], hit[1]


# Keyed on the point tensor OBJECT (identity + data_ptr + torch's version counter): a caller that refills the same tensor in
# place through raw pointers (which does not bump the counter) must call clear_voxelization_cache() at the fragment boundary;
# every path in this package hands over a fresh tensor per fragment.  Guarded by a lock: the pipelined serving mode calls in
# from a worker thread while the main thread runs the next fragment.
_VOX_CACHE = []
_VOX_CACHE_MAX = 6
_VOX_CACHE_LOCK = __import__("threading").Lock()


def clear_voxelization_cache():
    with _VOX_CACHE_LOCK:
        _VOX_CACHE.clear()


def _cached_entry(key, pts):
    with _VOX_CACHE_LOCK:
        for e in _VOX_CACHE:
            if e.key == key and e.pts is pts:
                return e
    return None


def _publish_entry(e):
    e.lists = _segment_lists(e.inverse, e.vset.n)
    e.idx8 = e.w8 = e._order = e.stride4 = None
    e._stale = {}
    with _VOX_CACHE_LOCK:
        _VOX_CACHE.append(e)
        if len(_VOX_CACHE) > _VOX_CACHE_MAX:
            _VOX_CACHE.pop(0)
    return e


def _voxelize_points(pts, res, levels=1):
    """levels > 1 (SPVCNN): the strided voxel sets of the U-Net's down stages are numbered in the same pass and ALL sizes are
    read back at once (sparse.voxel_hierarchy) — one host read per SPVCNN pass instead of three"""
    key = (pts.data_ptr(), pts._version, pts.shape[0], float(res))
    e = _cached_entry(key, pts)
    if e is not None:
        return e
    lib = _lib.load()
    n = pts.shape[0]
    e = _VoxEntry()
    e.key, e.pts = key, pts
    e.scaled = torch.empty_like(pts)
    e.vox = torch.empty((n, 4), dtype=torch.int32, device=pts.device)
    _lib.check(lib.eprecon_point_quantize_async(_lib.ptr(pts), n, float(res), _lib.ptr(e.scaled), _lib.ptr(e.vox),
                                                _lib.current_stream()), "eprecon_point_quantize_async")
    if levels > 1 and pts.is_cuda:
        e.vset, e.inverse, tables = SP.voxel_hierarchy(e.vox, levels, points=e.scaled)
        if tables is not None:
            # the pass's whole geometry came out of one library call: publish the entry complete
            e.lists = (tables["offsets1"], tables["order1"])
            e.idx8, e.w8, e._order, e._stale = tables["idx8_1"], tables["weight8_1"], None, {}
            e.stride4 = (tables["idx4"], (tables["offsets4"], tables["order4"]), tables["idx8_4"], tables["weight8_4"])
            with _VOX_CACHE_LOCK:
                _VOX_CACHE.append(e)
                del _VOX_CACHE[:max(0, len(_VOX_CACHE) - _VOX_CACHE_MAX)]
            return e
    else:
        uniq, e.inverse, grid = SP.unique_coords(e.vox, 1)
        e.vset = SP.VoxelSet(uniq, 1, grid=grid)
    return _publish_entry(e)


class SpvcnnPrefetch:
    """The coordinate side of the NEXT SPVCNN pass, queued while the length of its input list is still on the device.

    The voxels an SPVCNN pass runs on are the rows a compaction has just written — the occupied rows sparsify keeps
    (models/neucon_network.py:454-507, expanded to their 8 children :193-214) or the valid rows of the stage-0 back-projection —
    and the host learns their number from that compaction's count read.  The pass then needs a SECOND read: the sizes of its
    three strided voxel sets.  Here the chain [children ->] aligned-camera points -> scaled points / voxel indices -> unique
    numbering at strides 1, 2, 4 is queued on the device count (eprecon_spvcnn_points_dn_async, eprecon_unique_coords_dn_async)
    BEFORE the compaction's read; `headers()` are appended to that read and `finish()` slices the buffers and publishes the
    voxelisation under the points tensor it returns, where SPVCNN.forward's initial_voxelize finds it: one blocking read per
    level less, bit-identical results (the same kernels' arithmetic on the same rows)."""

    def __init__(self, src_coords, n_src_dev, children, interval, origin, voxel_size, world_to_aligned_camera, res, summary=None):
        """summary (optional): int32[6] device slice the six header words are written to (the tail of the compaction's own
        counts buffer: one tensor to read, no torch.cat)"""
        lib = _lib.load()
        dev = src_coords.device
        assert src_coords.dtype == torch.int32 and src_coords.is_contiguous()
        cap_src = src_coords.shape[0]
        cap = cap_src * 8 if children else cap_src
        origin = origin.float().reshape(-1, 3).contiguous()
        w2ac = world_to_aligned_camera.float().reshape(-1, 4, 4).contiguous()
        self.res = float(res)
        self.up = torch.empty((cap, 4), dtype=torch.int32, device=dev) if children else None
        self.r = torch.empty((cap, 4), dtype=torch.float32, device=dev)
        self.scaled = torch.empty((cap, 4), dtype=torch.float32, device=dev)
        self.vox = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        self.n_pts = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(lib.eprecon_spvcnn_points_dn_async(
            _lib.ptr(src_coords), cap_src, _lib.ptr(n_src_dev), int(bool(children)), int(interval), _lib.ptr(origin), origin.shape[0],
            float(voxel_size), _lib.ptr(w2ac), self.res, _lib.ptr(self.up), _lib.ptr(self.r), _lib.ptr(self.scaled),
            _lib.ptr(self.vox), _lib.ptr(self.n_pts), _lib.current_stream()), "eprecon_spvcnn_points_dn_async")
        self.summary = summary if summary is not None else torch.empty(6, dtype=torch.int32, device=dev)
        self.uniqs, self.invs, self.grids = SP.unique_hierarchy_queued(self.vox, 3, n_dev=self.n_pts, summary=self.summary)
        self._keep = (src_coords, n_src_dev, origin, w2ac)

    def headers(self):
        """int32[6]: (status word, unique count) of the three tables, to be read with the compaction's own counts"""
        return self.summary

    def finish(self, n_pts, host):
        """n_pts: live points (8 x kept rows, or the valid rows), host: the six header values -> (up_coords | None, r_coords):
        the rows of the new level and their aligned-camera points; the voxelisation (three strided sets + the pass's geometry)
        is published under `r_coords`"""
        sizes = []
        for lvl in range(3):
            SP.check_hash_status(host[2 * lvl])
            sizes.append(host[2 * lvl + 1])
        r = self.r[:n_pts]
        up = self.up[:n_pts] if self.up is not None else None
        if n_pts == 0 or sizes[0] == 0:
            return up, r
        e = _VoxEntry()
        e.key, e.pts = (r.data_ptr(), r._version, r.shape[0], self.res), r
        e.scaled, e.vox = self.scaled[:n_pts], self.vox[:n_pts]
        invs = [self.invs[0][:n_pts], self.invs[1], self.invs[2]]
        e.vset, e.inverse, tables = SP._hierarchy_with_geometry(e.vox, e.scaled, self.grids, self.uniqs, invs, sizes)
        e.lists = (tables["offsets1"], tables["order1"])
        e.idx8, e.w8, e._order, e._stale = tables["idx8_1"], tables["weight8_1"], None, {}
        e.stride4 = (tables["idx4"], (tables["offsets4"], tables["order4"]), tables["idx8_4"], tables["weight8_4"])
        with _VOX_CACHE_LOCK:
            _VOX_CACHE.append(e)
            del _VOX_CACHE[:max(0, len(_VOX_CACHE) - _VOX_CACHE_MAX)]
        return up, r


def register_voxelization(pts, res, scaled, vox, inverse, uniq, grid):
    """An entry whose quantisation and voxel numbering were computed elsewhere with device-side counts (GRU-fusion stage,
    eprecon_gru_stage_begin_async) and already sliced to their sizes: published under the points tensor `pts` so that the
    SConv3d that voxelise these points find it (initial_voxelize)"""
    e = _VoxEntry()
    e.key, e.pts = (pts.data_ptr(), pts._version, pts.shape[0], float(res)), pts
    e.scaled, e.vox, e.inverse = scaled, vox, inverse
    e.vset = SP.VoxelSet(uniq, 1, grid=grid)
    return _publish_entry(e)


def register_voxelization_pair(pts, res, first, second):
    """The two voxelisations the six SConv3d of a level's ConvGRUs share (of `pts` and of the once-scaled points), whose
    coordinate side came out of the GRU-fusion stage call: first / second = (scaled, vox, inverse, uniq, grid), already
    sliced to their sizes.  Everything else the cells need — CSR point lists, 3x3x3 kernel maps, corner tables, and in
    LITERAL_CONVR mode the reference's hash order + stale indices — is queued by ONE library call
    (eprecon_gru_stage_finish_async) and the two cache entries are published complete: the SConv3d layers only read."""
    lib = _lib.load()
    dev = pts.device
    n = pts.shape[0]
    (sc1, vox1, inv1, uq1, g1), (sc2, vox2, inv2, uq2, g2) = first, second
    m1, m2 = uq1.shape[0], uq2.shape[0]
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    d = _lib.GruFinishDesc()
    d.n, d.m1, d.m2 = n, m1, m2
    d.inverse1, d.inverse2, d.uniq1, d.uniq2 = inv1.data_ptr(), inv2.data_ptr(), uq1.data_ptr(), uq2.data_ptr()
    d.table1, d.table2, d.table_capacity = g1.mem.data_ptr(), g2.mem.data_ptr(), g1.capacity
    assert g1.capacity == g2.capacity
    d.scaled1, d.scaled2, d.literal = sc1.data_ptr(), sc2.data_ptr(), int(LITERAL_CONVR)
    off1, ord1, off2, ord2 = i32(m1 + 1), i32(max(n, 1)), i32(m2 + 1), i32(max(n, 1))
    nbr1, nbr2 = i32(27, m1), i32(27, m2)
    idx1, w1, idx2 = i32(n, 8), torch.empty((n, 8), dtype=torch.float32, device=dev), i32(n, 8)
    d.offsets1, d.order1, d.offsets2, d.order2 = off1.data_ptr(), ord1.data_ptr(), off2.data_ptr(), ord2.data_ptr()
    d.nbr1, d.nbr2, d.idx8_1, d.weight8_1, d.idx8_2 = nbr1.data_ptr(), nbr2.data_ptr(), idx1.data_ptr(), w1.data_ptr(), idx2.data_ptr()
    w2 = order = None
    if LITERAL_CONVR:
        order = i32(2, m1), i32(2, m2)
        d.perm1, d.rank1, d.perm2, d.rank2 = order[0][0].data_ptr(), order[0][1].data_ptr(), order[1][0].data_ptr(), order[1][1].data_ptr()
    else:
        w2 = torch.empty((n, 8), dtype=torch.float32, device=dev)
        d.weight8_2 = w2.data_ptr()
    ws = _lib.workspace(lib.eprecon_gru_stage_finish_workspace_bytes(n, m1, m2), dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(lib.eprecon_gru_stage_finish_async(ctypes.byref(d), _lib.current_stream()), "eprecon_gru_stage_finish_async")
    entries = []
    for src, scaled, vox, inv, uq, grid, lists, nbr in ((pts, sc1, vox1, inv1, uq1, g1, (off1, ord1), nbr1),
                                                         (sc1, sc2, vox2, inv2, uq2, g2, (off2, ord2), nbr2)):
        e = _VoxEntry()
        e.key, e.pts = (src.data_ptr(), src._version, src.shape[0], float(res)), src
        e.scaled, e.vox, e.inverse, e.lists = scaled, vox, inv, lists
        e.vset = SP.VoxelSet(uq, 1, grid=grid)
        e.vset._k3 = nbr
        e.idx8 = e.w8 = e._order = None
        e._stale = {}
        entries.append(e)
    e1, e2 = entries
    e1.idx8, e1.w8 = idx1, w1
    if LITERAL_CONVR:
        e1._order, e2._order = (order[0][0], order[0][1]), (order[1][0], order[1][1])
        e2._stale[id(e1)] = (idx2, w1, e1)
    else:
        e2.idx8, e2.w8 = idx2, w2
    with _VOX_CACHE_LOCK:
        _VOX_CACHE.extend(entries)
        del _VOX_CACHE[:max(0, len(_VOX_CACHE) - _VOX_CACHE_MAX)]
    return e1, e2


def _entry_corner_tables(e):
    """corner indices / weights of an entry's own points against its own voxel set (what the first voxel_to_point of
    an SConv3d on these points computes)"""
    if e.idx8 is None:
        lib = _lib.load()
        n = e.scaled.shape[0]
        idx8 = torch.empty((n, 8), dtype=torch.int32, device=e.scaled.device)
        w8 = torch.empty((n, 8), dtype=torch.float32, device=e.scaled.device)
        grid = e.vset.grid
        _lib.check(lib.eprecon_trilinear_map_async(_lib.ptr(grid.mem), grid.capacity, _lib.ptr(e.scaled), n, 1,
                                                   _lib.ptr(idx8), _lib.ptr(w8), _lib.current_stream()),
                   "eprecon_trilinear_map_async")
        e.idx8, e.w8 = idx8, w8
    return e.idx8, e.w8


def convgru_resolution(init_res, after_res):
    return float(after_res) / float(init_res) if init_res != 1 else float(after_res)


def prepare_convgru_voxelizations(coords, init_res, after_res):
    """Everything the six SConv3d of the two ConvGRUs of a scale share, built up front on the current stream: the
    voxelisation of `coords` (convz / convq) and of the once-scaled coordinates (convr), their kernel maps and corner
    tables (stale ones in LITERAL_CONVR mode).  Afterwards the two cells only READ these entries and can run on two
    streams.  (Entries registered by the GRU-fusion stage call are found in the cache: no quantise / unique / host read.)"""
    res = convgru_resolution(init_res, after_res)
    pts = coords if coords.is_contiguous() else coords.contiguous()
    e1 = _voxelize_points(pts, res)
    e1.vset.kernel_map(3)
    _entry_corner_tables(e1)
    e2 = _voxelize_points(e1.scaled, res)
    e2.vset.kernel_map(3)
    if LITERAL_CONVR:
        e2.stale_from(e1)
    else:
        _entry_corner_tables(e2)
    return e1, e2


def initial_voxelize(z, init_res, after_res, levels=1):
    """ops/torchsparse_utils.py:15-35: floor(z.C * init_res / after_res) -> unique voxels
    (first-occurrence order) -> scatter-mean of z.F.  Overwrites z.C with the scaled coordinates.

    A second call on the same PointTensor (ConvGRU: convz then convr on `hx`, models/modules.py:214-217)
    voxelises the already-scaled coordinates, and — like the reference, whose voxel_to_point finds
    z.idx_query[1] / z.weights[1] filled by the first call (ops/torchsparse_utils.py:70-71,97-99) — keeps
    devoxelising with the FIRST call's corner indices and weights, now pointing into the second voxel set
    in torchsparse's ascending-hash voxel order (LITERAL_CONVR)."""
    pts = z.C if z.C.is_contiguous() else z.C.contiguous()
    res = float(after_res) / float(init_res) if init_res != 1 else float(after_res)
    prev = getattr(z, "_vox_entry", None)
    e = _voxelize_points(pts, res, levels)
    feat = _segment_mean(z.F, e.lists, e.vset.n, idx=e.inverse)
    z.C, z.vox = e.scaled, e.vox
    if prev is not None and LITERAL_CONVR and 1 in z.idx_query:
        z.idx_query[1], z.weights[1] = e.stale_from(prev)
    else:
        z.idx_query.clear()
        z.weights.clear()
        if e.idx8 is not None:
            z.idx_query[1], z.weights[1] = e.idx8, e.w8
    z.additional_features["idx_query"].clear()
    z.additional_features["lists"].clear()
    z.additional_features["idx_query"][1] = e.inverse
    z.additional_features["lists"][1] = e.lists
    s4 = getattr(e, "stride4", None)
    if s4 is not None and not (prev is not None and LITERAL_CONVR):
        # (SPVCNN: the stride-4 transfers of the pass were tabulated with the rest of its geometry)
        z.additional_features["idx_query"][4], z.additional_features["lists"][4] = s4[0], s4[1]
        z.idx_query[4], z.weights[4] = s4[2], s4[3]
    z._vox_entry = e
    return SparseTensor(feat, e.vset)


def point_to_voxel(x, z, out=None):
    """ops/torchsparse_utils.py:40-63: scatter-mean of z.F into the voxels of x (tensor stride x.s)"""
    s = x.s
    lists = z.additional_features["lists"].get(s)
    if lists is None:
        idx = x.vset.grid.query(z.vox, quantum=s)
        lists = _segment_lists(idx, x.vset.n)
        z.additional_features["idx_query"][s] = idx
        z.additional_features["lists"][s] = lists
    return SparseTensor(_segment_mean(z.F, lists, x.vset.n, out, idx=z.additional_features["idx_query"][s]), x.vset)


def _corner_tables(vset, s, z):
    """8-corner indices / renormalised trilinear weights of the points of z against `vset` at tensor stride s,
    cached on the PointTensor like the reference's z.idx_query / z.weights (ops/torchsparse_utils.py:70-96)"""
    if s not in z.idx_query:
        lib = _lib.load()
        n = z.C.shape[0]
        idx8 = torch.empty((n, 8), dtype=torch.int32, device=z.C.device)
        w8 = torch.empty((n, 8), dtype=torch.float32, device=z.C.device)
        grid = vset.grid
        _lib.check(lib.eprecon_trilinear_map_async(_lib.ptr(grid.mem), grid.capacity, _lib.ptr(z.C), n, s,
                                                   _lib.ptr(idx8), _lib.ptr(w8), _lib.current_stream()),
                   "eprecon_trilinear_map_async")
        z.idx_query[s], z.weights[s] = idx8, w8
        e = getattr(z, "_vox_entry", None)
        if s == 1 and e is not None and e.vset is vset and e.idx8 is None:
            e.idx8, e.w8 = idx8, w8   # shared with the other SConv3d that voxelise the same points
    return z.idx_query[s], z.weights[s]


def voxel_to_point(x, z, nearest=False, out=None, accumulate=False):
    """ops/torchsparse_utils.py:68-105: trilinear interpolation of the voxel features of x at the
    points of z (weights renormalised over the corners that exist)."""
    assert not nearest  # never used with True by the reference
    lib = _lib.load()
    s = x.s
    n = z.C.shape[0]
    _corner_tables(x.vset, s, z)
    c = x.F.shape[1]
    if torch.is_grad_enabled() and (x.F.requires_grad or (out is not None and out.requires_grad)):
        res = AG.devoxelize(x.F, z.idx_query[s], z.weights[s])
        out = res if out is None else (out + res if accumulate else out.copy_(res))
    else:
        if out is None:
            out = torch.empty((n, c), dtype=torch.float32, device=x.F.device)
        _lib.check(lib.eprecon_devoxelize_async(_lib.ptr(x.F), x.F.stride(0), _lib.ptr(z.idx_query[s]),
                                                _lib.ptr(z.weights[s]), n, c, _lib.ptr(out), out.stride(0),
                                                int(accumulate), _lib.current_stream()), "eprecon_devoxelize_async")
    new = PointTensor(out, z.C, idx_query=z.idx_query, weights=z.weights)
    new.vox = z.vox
    new.additional_features = z.additional_features
    new._vox_entry = getattr(z, "_vox_entry", None)
    return new


def devoxelize_gate(x, z, skip, mode, h=None, zgate=None, out=None, tail=None):
    """voxel_to_point(x, z).F + skip followed by the ConvGRU gate arithmetic, one launch
    (eprecon_devoxelize_gate_async; mode 1 sigmoid, 2 sigmoid * h, 3 (1 - zgate) * h + zgate * tanh).
    tail = (src, dst): row-wise copy dst[:] = src[:] of two [n, t] column slices in the same launch"""
    lib = _lib.load()
    s = x.s
    n, c = z.C.shape[0], x.F.shape[1]
    _corner_tables(x.vset, s, z)
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.F.device)
    t_src, t_dst = tail if tail is not None else (None, None)
    _lib.check(lib.eprecon_devoxelize_gate_tail_async(
        _lib.ptr(x.F), x.F.stride(0), _lib.ptr(z.idx_query[s]), _lib.ptr(z.weights[s]), n, c, _lib.ptr(skip),
        skip.stride(0), int(mode), _lib.ptr(h), h.stride(0) if h is not None else 0, _lib.ptr(zgate),
        zgate.stride(0) if zgate is not None else 0, _lib.ptr(out), out.stride(0), _lib.ptr(t_src),
        t_src.stride(0) if t_src is not None else 0, _lib.ptr(t_dst), t_dst.stride(0) if t_dst is not None else 0,
        t_src.shape[1] if t_src is not None else 0, _lib.current_stream()), "eprecon_devoxelize_gate_tail_async")
    return out
