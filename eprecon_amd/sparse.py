"""Host-side handles for the sparse primitives of libeprecon_hip.so: coordinate sets with their
hash grid and cached kernel maps (the role torchsparse's SparseTensor.cmaps/.kmaps and spconv's
indice pairs play in the reference), sparse convolution, and the normalisation epilogues.

Coordinates are int32[N,4] rows (batch, x, y, z) everywhere in this package; wrappers that mirror
torchsparse's xyzb order convert at their boundary.
"""
import ctypes

import torch

from . import _lib


import os

def _dense3d_level():
    """EPRECON_CONV_DENSE3D: 0 off, 1 the single-column (C_out == 1) kernel only, 2 (default) also the 16-row MFMA tile kernel
    (C_out <= 32, C_in % 16 == 0); csrc/sparse_conv.hip, conv3d_kind"""
    return int(os.environ.get("EPRECON_CONV_DENSE3D", "2"))


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D tensor expected"
    return t.stride(0)


class HashGrid:
    """device hash table over a coordinate set (eprecon_hash_*)"""

    def __init__(self, n, device):
        lib = _lib.load()
        self.capacity = int(lib.eprecon_hash_capacity(int(n)))
        self.mem = torch.empty(int(lib.eprecon_hash_table_bytes(self.capacity)), dtype=torch.uint8,
                               device=device)

    @property
    def header(self):
        """int32[2] view of the table's header: (status word, count of unique keys written by the unique calls)"""
        return self.mem[:8].view(torch.int32)

    def build(self, coords, quantum=1, n_dev=None):
        """n_dev (a device int32 element): only the first min(N, *n_dev) rows of `coords` are live"""
        lib = _lib.load()
        if n_dev is not None:
            _lib.check(lib.eprecon_hash_build_dn_async(_lib.ptr(coords), coords.shape[0], _lib.ptr(n_dev), quantum,
                                                       _lib.ptr(self.mem), self.capacity, _lib.current_stream()),
                       "eprecon_hash_build_dn_async")
            return self
        _lib.check(lib.eprecon_hash_build_async(_lib.ptr(coords), coords.shape[0], quantum,
                                                _lib.ptr(self.mem), self.capacity,
                                                _lib.current_stream()), "eprecon_hash_build_async")
        return self

    def query(self, queries, quantum=1):
        lib = _lib.load()
        out = torch.empty(queries.shape[0], dtype=torch.int32, device=queries.device)
        _lib.check(lib.eprecon_hash_query_async(_lib.ptr(self.mem), self.capacity, _lib.ptr(queries),
                                                queries.shape[0], quantum, _lib.ptr(out),
                                                _lib.current_stream()), "eprecon_hash_query_async")
        return out

    def status_ok(self):
        """blocking; raises when a key was out of range or the table overflowed"""
        lib = _lib.load()
        return _lib.check(lib.eprecon_hash_status(_lib.ptr(self.mem), _lib.current_stream()),
                          "eprecon_hash_status")


def check_hash_status(status):
    """status word of a hash table (bit 0: a key out of the packable range — |coordinate| >= 2^19 - 1 or batch > 14 —, bit 1:
    table full): such voxels would otherwise silently drop out of every kernel map built on the set"""
    if status:
        raise _lib.EpreconError(f"hash grid: {'coordinate / batch index out of range' if status & 1 else 'table full'} "
                                f"(status {status}, EPRECON_ERR_UNSUPPORTED)")


def unique_coords_queued(coords, quantum=1, n_dev=None):
    """Queue the unique (quantised) voxel numbering of `coords` WITHOUT reading the count back:
    -> (unique int32[N,4] whose first M rows are written, inverse int32[N], HashGrid).  M and the table's status word land in
    grid.header (device); n_dev (a device int32 element): only the first min(N, *n_dev) rows of `coords` are live."""
    lib = _lib.load()
    coords = coords.contiguous()
    n, dev = coords.shape[0], coords.device
    grid = HashGrid(n, dev)
    inverse = torch.empty(n, dtype=torch.int32, device=dev)
    uniq = torch.empty((n, 4), dtype=torch.int32, device=dev)
    n_unique = grid.header[1:2]
    ws = _lib.workspace(lib.eprecon_unique_workspace_bytes(n), dev)
    if n_dev is None:
        _lib.check(lib.eprecon_unique_coords_async(_lib.ptr(coords), n, quantum, _lib.ptr(grid.mem), grid.capacity,
                                                   _lib.ptr(inverse), _lib.ptr(uniq), _lib.ptr(n_unique), _lib.ptr(ws), ws.numel(),
                                                   _lib.current_stream()), "eprecon_unique_coords_async")
    else:
        _lib.check(lib.eprecon_unique_coords_dn_async(_lib.ptr(coords), n, _lib.ptr(n_dev), quantum, _lib.ptr(grid.mem),
                                                      grid.capacity, _lib.ptr(inverse), _lib.ptr(uniq), _lib.ptr(n_unique),
                                                      _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                   "eprecon_unique_coords_dn_async")
    return uniq, inverse, grid


def unique_hierarchy_queued(coords, levels=3, n_dev=None, summary=None):
    """unique_coords_queued for the strides 1, 2, 4 (levels <= 3) of one point cloud in ONE library call (the tables of all strides are
    reset by one launch; level l numbers the unique rows of level l - 1 at quantum 2^l, its count taken from the device):
    -> (uniqs, inverses, grids), each a list over the levels; the counts / status words sit in grids[l].header and, side by
    side, in `summary` (an int32[2 levels] device tensor of the caller's: what it reads back, no torch.cat of the headers)"""
    if not 1 <= levels <= 3:    # (the library call numbers at most three strides and would answer EPRECON_ERR_ARG)
        raise ValueError(f"unique_hierarchy_queued: levels must be 1, 2 or 3 (strides 1, 2, 4), got {levels}")
    lib = _lib.load()
    coords = coords.contiguous()
    n, dev = coords.shape[0], coords.device
    grids = [HashGrid(n, dev) for _ in range(levels)]
    invs = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(levels)]
    uniqs = [torch.empty((n, 4), dtype=torch.int32, device=dev) for _ in range(levels)]
    ws = _lib.workspace(lib.eprecon_unique_workspace_bytes(n), dev)
    vp = ctypes.c_void_p * levels
    tables = vp(*[g.mem.data_ptr() for g in grids])
    caps = (ctypes.c_uint32 * levels)(*[g.capacity for g in grids])
    inv_p, uniq_p = vp(*[t.data_ptr() for t in invs]), vp(*[t.data_ptr() for t in uniqs])
    _lib.check(lib.eprecon_unique_hierarchy_dn_async(_lib.ptr(coords), n, _lib.ptr(n_dev), levels, tables, caps, inv_p, uniq_p,
                                                     _lib.ptr(summary), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "eprecon_unique_hierarchy_dn_async")
    return uniqs, invs, grids


def unique_coords(coords, quantum=1):
    """-> (unique int32[M,4] in first-occurrence order, inverse int32[N], HashGrid mapping key -> id).
    One host sync to learn M (the reference's torch.unique syncs as well)."""
    uniq, inverse, grid = unique_coords_queued(coords, quantum)
    # the count lands next to the table's status word (the table's 256-byte header): ONE 8-byte host read, no torch.cat
    status, m = _lib.read_counts(grid.header)
    check_hash_status(status)
    return uniq[:m], inverse, grid


DIRECT_MAX_COUT = 64   # (operand-order packings are always handed over; which kernel runs is the library's choice)
DENSE_MIN_FILL = 0.4   # a set that fills at least this share of its bounding grid takes the dense-grid convolution


class DenseMap:
    """What a 3x3x3 stride-1 convolution needs on a voxel set that lives on a dense grid: the rank volume
    int32[gx*gy*gz (+1)] (cell -> row or -1, eprecon_grid_rank_async) instead of a hash grid + [27, N] kernel map.
    Layers whose shape the tile kernel does not take fall back to the set's kernel map (built on first use)."""

    def __init__(self, vset, dims):
        lib = _lib.load()
        self.vset, self.dims = vset, tuple(int(d) for d in dims)
        gx, gy, gz = self.dims
        self.rank = torch.empty(gx * gy * gz + 1, dtype=torch.int32, device=vset.coords.device)
        _lib.check(lib.eprecon_grid_rank_async(_lib.ptr(vset.coords), vset.n, vset.stride, gx, gy, gz, _lib.ptr(self.rank),
                                               _lib.current_stream()), "eprecon_grid_rank_async")
        # rank[-1] counts the voxels that were NOT on the declared grid (their output rows would stay unwritten): checked with
        # the next blocking count read on this stream (_lib.read_counts), at no read of its own
        _lib.defer_check(self.rank[-1:], 0, "dense-grid convolution: voxels of the set are not on the grid their VoxelSet was "
                                            "declared with (rows left unwritten)")

    @property
    def shape(self):          # (K, N) like the kernel-map tensor it stands in for
        return (27, self.vset.n)

    def off_grid(self):
        """blocking: number of voxels that were not on the grid (0 for a valid set)"""
        return int(self.rank[-1].item())

    def kind(self, x, cin, cout, accumulate=False, ln=False, stats=False):
        """mirror of the library's rule (conv3d_kind, csrc/sparse_conv.hip): 0 none (kernel map), 1 single-column kernel,
        2 16-row MFMA kernel"""
        level = _dense3d_level()
        if level <= 0 or cin % 4 or cin > 64 or x.stride(0) % 4 or x.data_ptr() % 16:
            return 0
        if cout == 1 and not ln:
            return 1
        if level < 2 or accumulate:
            return 0
        if cout <= 32 and cin % 16 == 0 and not (ln and stats):
            return 2
        return 0

    def takes(self, x, cin, cout, accumulate=False, ln=False, stats=False):
        return self.kind(x, cin, cout, accumulate, ln, stats) != 0


# Every packing made from a weight tensor's own storage is remembered here (weak reference to the tensor object the cache lives
# on): repack_registered() rebuilds them all in ONE launch after an optimizer step (TrainStep.run) instead of one launch per
# layer at its next use — ~190 launches per step.  id(owner) -> {kind: (weakref, shape)}
_PACK_REGISTRY = {}
_PACK_TABLE = {"key": None, "jobs": None, "entries": None}


def _register_pack(owner, kind, shape, source_ptr):
    import weakref
    if source_ptr != owner.data_ptr():      # packed from a temporary contiguous copy: nothing a later launch could re-read
        return
    slot = _PACK_REGISTRY.setdefault(id(owner), {})
    if kind not in slot or slot[kind][0]() is not owner or slot[kind][1] != tuple(shape):
        slot[kind] = (weakref.ref(owner), tuple(shape))
        _PACK_TABLE["key"] = None


def repack_registered():
    """Rebuild every registered operand-order copy whose weight has changed since it was packed (an optimizer step changes
    them all), in one launch on the current stream; the caches' tags follow, so the layers' next packed_weight /
    packed_weight16 calls are hits.  Tensors that were freed, moved or re-shaped drop out of the registry.  -> jobs launched"""
    import numpy as np
    entries = []
    for oid in list(_PACK_REGISTRY):
        slot = _PACK_REGISTRY[oid]
        for kind in list(slot):
            ref, shape = slot[kind]
            owner = ref()
            hit = getattr(owner, "_d3_pack" if kind == 0 else "_d3_pack16", None) if owner is not None else None
            if owner is None or hit is None or not owner.is_cuda or (kind == 1 and hit[0][2] != shape):
                del slot[kind]
                _PACK_TABLE["key"] = None
                continue
            entries.append((owner, kind, shape, hit[1]))
        if not slot:
            del _PACK_REGISTRY[oid]
    stale = [e for e in entries if (e[0]._version, e[0].data_ptr()) != getattr(e[0], "_d3_pack" if e[1] == 0 else "_d3_pack16")[0][:2]]
    if not stale:
        return 0
    key = tuple((o.data_ptr(), k, s, p.data_ptr()) for o, k, s, p in stale)
    if _PACK_TABLE["key"] != key:     # (the table is uploaded once: pointers and shapes do not change from step to step)
        job = np.zeros(len(stale), dtype=np.dtype([("weight", "<u8"), ("packed", "<u8"), ("kvol", "<i4"), ("cin", "<i4"),
                                                   ("cout", "<i4"), ("kind", "<i4")]))
        for i, (o, k, s, p) in enumerate(stale):
            job[i] = (o.data_ptr(), p.data_ptr(), s[0], s[1], s[2], k)
        _PACK_TABLE["jobs"] = torch.from_numpy(job.view(np.uint8).copy()).to(stale[0][0].device)
        _PACK_TABLE["key"] = key
    lib = _lib.load()
    _lib.check(lib.eprecon_conv_pack_many_async(_lib.ptr(_PACK_TABLE["jobs"]), len(stale), _lib.current_stream()),
               "eprecon_conv_pack_many_async")
    for o, k, s, p in stale:
        if k == 0:
            o._d3_pack = ((o._version, o.data_ptr()), p)
        else:
            o._d3_pack16 = ((o._version, o.data_ptr(), s), p)
    return len(stale)


def packed_weight(weight):
    """`weight` f32[27, Cin, Cout] in the operand order of the dense-grid kernel, packed once per weight version"""
    hit = getattr(weight, "_d3_pack", None)
    tag = (weight._version, weight.data_ptr())
    if hit is None or hit[0] != tag or hit[1].device != weight.device:
        lib = _lib.load()
        kvol, cin, cout = weight.shape
        w = weight.detach().contiguous()
        packed = torch.empty(int(lib.eprecon_conv_pack_weight_floats(kvol, cin, cout)), dtype=torch.float32, device=weight.device)
        _lib.check(lib.eprecon_conv_pack_weight_async(_lib.ptr(w), kvol, cin, cout, _lib.ptr(packed), _lib.current_stream()),
                   "eprecon_conv_pack_weight_async")
        hit = (tag, packed)
        weight._d3_pack = hit
        _register_pack(weight, 0, (kvol, cin, cout), w.data_ptr())
    return hit[1]


def packed_weight16(weight, owner=None):
    """`weight` f32[K, Cin, Cout <= 64] in the operand order of the 16x16x4 MFMA kernels (16-row tile kernel, direct gather
    kernel), packed once per weight version.  owner: the tensor OBJECT the packing is cached on when `weight` is a transient
    view of it (a [Cin, Cout] matrix unsqueezed to K = 1)"""
    owner = weight if owner is None else owner
    hit = getattr(owner, "_d3_pack16", None)
    tag = (owner._version, owner.data_ptr(), tuple(weight.shape))
    if hit is None or hit[0] != tag or hit[1].device != weight.device:
        lib = _lib.load()
        kvol, cin, cout = weight.shape
        w = weight.detach().contiguous()
        packed = torch.empty(int(lib.eprecon_conv_pack_weight16_floats(kvol, cin, cout)), dtype=torch.float32, device=weight.device)
        _lib.check(lib.eprecon_conv_pack_weight16_async(_lib.ptr(w), kvol, cin, cout, _lib.ptr(packed), _lib.current_stream()),
                   "eprecon_conv_pack_weight16_async")
        hit = (tag, packed)
        owner._d3_pack16 = hit
        _register_pack(owner, 1, (kvol, cin, cout), w.data_ptr())
    return hit[1]


def voxel_hierarchy(vox, levels=3, points=None):
    """The voxel sets of a point cloud at tensor strides 1, 2, 4, ... with ONE host read for all their sizes: the unique
    numbering of `vox` int32[N,4] and of each coarser stride is queued back to back (the coarser calls take the finer
    count from the device, eprecon_unique_coords_dn_async), then the `levels` counts and status words are read together.
    -> (VoxelSet at stride 1 with its downsample chain attached, inverse int32[N]).
    points = scaled f32[N,4] (levels == 3, SPVCNN): everything else a pass needs from the coordinates — CSR point lists and
    trilinear corner tables of strides 1 and 4, the strided maps, the three kernel maps — is queued by ONE library call
    behind the read (eprecon_spvcnn_geometry_async) -> (VoxelSet, inverse, tables)."""
    n = vox.shape[0]
    summary = torch.empty(2 * levels, dtype=torch.int32, device=vox.device)
    uniqs, invs, grids = unique_hierarchy_queued(vox, levels, summary=summary)
    host = _lib.read_counts(summary)
    sizes = []
    for lvl in range(levels):
        check_hash_status(host[2 * lvl])
        sizes.append(host[2 * lvl + 1])
    if points is not None and levels == 3 and n > 0 and sizes[0] > 0:
        return _hierarchy_with_geometry(vox, points, grids, uniqs, invs, sizes)
    base = VoxelSet(uniqs[0][:sizes[0]], 1, grid=grids[0])
    cur = base
    for lvl in range(1, levels):
        coarse, _, _ = cur.downsample(pre=(uniqs[lvl][:sizes[lvl]], invs[lvl][:sizes[lvl - 1]], grids[lvl]))
        cur = coarse
    return (base, invs[0]) if points is None else (base, invs[0], None)


def _hierarchy_with_geometry(vox, points, grids, uniqs, invs, sizes):
    lib = _lib.load()
    dev = vox.device
    n = vox.shape[0]
    n1, n2, n4 = sizes
    c1, c2, c4 = uniqs[0][:n1], uniqs[1][:n2], uniqs[2][:n4]
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    t = {"offsets1": i32(n1 + 1), "order1": i32(n), "idx4": i32(n), "offsets4": i32(n4 + 1), "order4": i32(n),
         "down12": i32(8, n2), "up21": i32(8, n1), "down24": i32(8, n4), "up42": i32(8, n2),
         "k1": i32(27, n1), "k2": i32(27, n2), "k4": i32(27, n4), "idx8_1": i32(n, 8), "idx8_4": i32(n, 8),
         "weight8_1": torch.empty((n, 8), dtype=torch.float32, device=dev),
         "weight8_4": torch.empty((n, 8), dtype=torch.float32, device=dev)}
    d = _lib.SpvcnnGeometryDesc()
    d.n, d.n1, d.n2, d.n4 = n, n1, n2, n4
    d.scaled, d.vox, d.inverse1 = points.data_ptr(), vox.data_ptr(), invs[0].data_ptr()
    d.coords1, d.coords2, d.coords4 = c1.data_ptr(), c2.data_ptr(), c4.data_ptr()
    d.parent2, d.parent4 = invs[1].data_ptr(), invs[2].data_ptr()
    d.table1, d.table2, d.table4 = (g.mem.data_ptr() for g in grids)
    d.capacity1, d.capacity2, d.capacity4 = (g.capacity for g in grids)
    for name, buf in t.items():
        setattr(d, name, buf.data_ptr())
    ws = _lib.workspace(lib.eprecon_spvcnn_geometry_workspace_bytes(n, n1, n4), dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(lib.eprecon_spvcnn_geometry_async(ctypes.byref(d), _lib.current_stream()), "eprecon_spvcnn_geometry_async")
    s1, s2, s4 = VoxelSet(c1, 1, grid=grids[0]), VoxelSet(c2, 2, grid=grids[1]), VoxelSet(c4, 4, grid=grids[2])
    s1._k3, s2._k3, s4._k3 = t["k1"], t["k2"], t["k4"]
    s1._down, s2._down = (s2, t["down12"], t["up21"]), (s4, t["down24"], t["up42"])
    return s1, invs[0], t


def clear_packed_weights(module):
    """Drop every operand-order copy cached on the parameters / layers of `module` (packed_weight, packed_weight16,
    dense2d.packed_weight, modules._linear_wt, SPVCNN's native-pass descriptor).  The caches are keyed on (tensor version, data_ptr): writes through `p.data`
    (dist.broadcast(p.data, ...), EMA swaps, manual loaders) change neither — call this after them."""
    for p_ in module.parameters():
        for attr in ("_d3_pack", "_d3_pack16"):
            if hasattr(p_, attr):
                delattr(p_, attr)
    for m in module.modules():
        # (_native / _native_params: SPVCNN's descriptor of the one-call pass, which holds packed copies of all of the above)
        for attr in ("_wt_cache", "_eprecon_packed", "_eprecon_merged", "_native", "_native_params"):
            if hasattr(m, attr):
                delattr(m, attr)


# point-wise layers (K = 1) on lists at least this long take the direct kernel too (a streaming [N, C_in] x [C_in, C_out <= 64]
# product: csrc/sparse_conv_direct_impl.hpp); shorter lists are launch-bound on any kernel
K1_DIRECT_MIN_ROWS = 20000


def _resolve_map(nbr, x, weight, desc, accumulate=False, ln=False, stats=False, owner=None):
    """nbr: None (identity), an int32[K, N] kernel map, or a DenseMap -> fills the map fields of `desc`; returns the
    objects that must stay alive until the launch is queued.  owner: the caller's weight object (packings are cached on it)"""
    if isinstance(nbr, DenseMap):
        kvol, cin, cout = weight.shape
        kind = nbr.kind(x, cin, cout, accumulate, ln, stats) if kvol == 27 else 0
        if kind:
            desc.vox_rank = nbr.rank.data_ptr()
            desc.grid_x, desc.grid_y, desc.grid_z = nbr.dims
            keep = [nbr.rank]
            if kind == 2:
                pw = packed_weight16(weight)
                desc.packed_weight16 = pw.data_ptr()
                keep.append(pw)
            desc.nbr = None
            return keep
        nbr = nbr.vset.kernel_map(3)
    desc.nbr = None if nbr is None else nbr.data_ptr()
    keep = [nbr]
    if nbr is None and weight.shape[0] == 1 and weight.shape[2] <= DIRECT_MAX_COUT and not accumulate and x.is_cuda \
            and x.shape[0] >= K1_DIRECT_MIN_ROWS:
        pw = packed_weight16(weight, owner)
        desc.packed_weight16 = pw.data_ptr()
        keep.append(pw)
    if nbr is not None and weight.shape[0] == 27 and weight.shape[2] <= DIRECT_MAX_COUT \
            and not accumulate and x.is_cuda:
        # long lists: the direct gather kernel takes its B operands pre-packed (csrc/sparse_conv.hip, spconv_direct16_kernel);
        # which kernel runs is the library's choice, the packing only makes the direct one possible
        pw = packed_weight16(weight)
        desc.packed_weight16 = pw.data_ptr()
        keep.append(pw)
    if nbr is not None and weight.shape[0] == 27 and not accumulate and x.is_cuda:
        # short lists: the split-K kernel reads its B operands straight from the 32x32x2 operand-order packing
        pw = packed_weight(weight)
        desc.packed_weight = pw.data_ptr()
        keep.append(pw)
    return keep


class VoxelSet:
    """A set of active voxels at one tensor stride: coords int32[N,4] (b,x,y,z), its hash grid and
    the kernel maps built on it.  Maps are built once and reused by every layer on the set.
    `dims` (optional): the set lives on the dense grid of dims cells of `stride` voxels starting at 0 (a raster of
    generate_grid, ops/generate_grids.py:3-10) — its 3x3x3 layers then run on the dense-grid kernel when it is full enough."""

    def __init__(self, coords, stride=1, grid=None, dims=None):
        assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
        self.coords = coords.contiguous()
        self.stride = int(stride)
        self.dims = None if dims is None else tuple(int(d) for d in dims)
        self._grid = grid
        self._k3 = None
        self._down = None
        self._dense = None

    def conv_map(self, ksize=3):
        """what a stride-1 k=3 layer on this set takes as its map: a DenseMap for a well-filled dense grid, else the
        [27, N] kernel map"""
        assert ksize == 3
        if self.dims is not None and self.n >= DENSE_MIN_FILL * self.dims[0] * self.dims[1] * self.dims[2] \
                and _dense3d_level() > 0 and self.coords.is_cuda:
            if self._dense is None:
                self._dense = DenseMap(self, self.dims)
            return self._dense
        return self.kernel_map(3)

    @property
    def n(self):
        return self.coords.shape[0]

    @property
    def grid(self):
        if self._grid is None:
            self._grid = HashGrid(self.n, self.coords.device).build(self.coords)
        return self._grid

    def kernel_map(self, ksize=3):
        """int32[27, N] neighbour table of the stride-1 k=3 convolution on this set"""
        assert ksize == 3
        if self._k3 is None:
            lib = _lib.load()
            nbr = torch.empty((27, self.n), dtype=torch.int32, device=self.coords.device)
            _lib.check(lib.eprecon_kernel_map_async(_lib.ptr(self.grid.mem), self.grid.capacity,
                                                    _lib.ptr(self.coords), self.n, 3, self.stride,
                                                    _lib.ptr(nbr), _lib.current_stream()),
                       "eprecon_kernel_map_async")
            self._k3 = nbr
        return self._k3

    def downsample(self, pre=None):
        """k2s2 strided set: -> (coarse VoxelSet, down_map int32[8, M], up_map int32[8, N]).
        pre = (unique int32[M,4], parent int32[N], HashGrid): the strided set numbered by an earlier queued call
        (voxel_hierarchy: the counts of all strides read back at once)"""
        if self._down is None:
            lib = _lib.load()
            q = 2 * self.stride
            uniq, parent, cgrid = pre if pre is not None else unique_coords(self.coords, quantum=q)
            coarse = VoxelSet(uniq, q, grid=cgrid)
            m = coarse.n
            down = torch.empty((8, m), dtype=torch.int32, device=self.coords.device)
            _lib.check(lib.eprecon_kernel_map_async(_lib.ptr(self.grid.mem), self.grid.capacity,
                                                    _lib.ptr(coarse.coords), m, 2, self.stride,
                                                    _lib.ptr(down), _lib.current_stream()),
                       "eprecon_kernel_map_async")
            up = torch.empty((8, self.n), dtype=torch.int32, device=self.coords.device)
            _lib.check(lib.eprecon_transpose_map_async(_lib.ptr(self.coords), self.n, _lib.ptr(parent),
                                                       self.stride, _lib.ptr(up), _lib.current_stream()),
                       "eprecon_transpose_map_async")
            self._down = (coarse, down, up)
        return self._down


def _attach_workspace(desc, device):
    """scratch for the kernels that reduce partial sums across workgroups (medium lists with wide channels): the library says
    how much the described launch wants (0 for every other shape); returns the buffer to keep alive until the launch is queued"""
    need = int(_lib.load().eprecon_conv_desc_workspace_bytes(ctypes.byref(desc)))
    if not need:
        return None
    ws = _lib.workspace(need, device)
    desc.workspace, desc.workspace_bytes = ws.data_ptr(), ws.numel()
    return ws


def sparse_conv(x, weight, nbr=None, bias=None, out=None, relu=False, accumulate=False):
    """out[i] (op)= bias + sum_k x[nbr[k][i]] @ weight[k].  weight f32[K, Cin, Cout] (or [Cin, Cout]
    with nbr None: a per-voxel linear layer).  x / out may be column slices of wider buffers."""
    lib = _lib.load()
    w_owner = weight
    if weight.dim() == 2:
        weight = weight.unsqueeze(0)
    kvol, cin, cout = weight.shape
    weight = weight.contiguous()
    n_out = x.shape[0] if nbr is None else nbr.shape[1]
    assert x.shape[1] == cin and x.dtype == torch.float32
    if isinstance(nbr, DenseMap) or (kvol == 27 and nbr is not None and x.is_cuda) or \
            (kvol == 1 and nbr is None and x.is_cuda and cout <= DIRECT_MAX_COUT and not accumulate and n_out >= K1_DIRECT_MIN_ROWS):
        return sparse_conv_fused(x, w_owner, nbr, bias, out, relu, None, accumulate)[0]
    if nbr is not None:
        assert nbr.dtype == torch.int32 and nbr.shape[0] == kvol and nbr.is_contiguous()
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    assert out.shape == (n_out, cout)
    _lib.check(lib.eprecon_sparse_conv_async(_lib.ptr(x), x.shape[0], _ld(x), _lib.ptr(nbr), kvol, n_out,
                                             _lib.ptr(weight), cin, cout, _lib.ptr(bias), _lib.ptr(out),
                                             _ld(out), int(relu), int(accumulate), _lib.current_stream()),
               "eprecon_sparse_conv_async")
    return out


def sparse_conv_fused(x, weight, nbr=None, bias=None, out=None, relu=False, residual=None, accumulate=False,
                      bn_partial=False):
    """sparse_conv with the fused epilogues: v = conv + bias [+ out]; [relu]; [+ residual]; and, with
    bn_partial, the per-workgroup BatchNorm summaries f32[ceil(n/128), 3, cout] of the stored values.
    Returns (out, partial or None)."""
    lib = _lib.load()
    w_owner = weight
    if weight.dim() == 2:
        weight = weight.unsqueeze(0)
    kvol, cin, cout = weight.shape
    weight = weight.contiguous()
    n_out = x.shape[0] if nbr is None else nbr.shape[1]
    assert x.shape[1] == cin and x.dtype == torch.float32
    k1_direct = kvol == 1 and nbr is None and x.is_cuda and cout <= DIRECT_MAX_COUT and not accumulate and n_out >= K1_DIRECT_MIN_ROWS
    if isinstance(nbr, DenseMap) or (kvol == 27 and nbr is not None and x.is_cuda) or k1_direct:   # descriptor entry point
        if nbr is not None and not isinstance(nbr, DenseMap):
            assert nbr.dtype == torch.int32 and nbr.shape[0] == kvol and nbr.is_contiguous()
        if out is None:
            out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
        assert out.shape == (n_out, cout)
        d = _lib.ConvDesc()
        d.x, d.n_in, d.ld_x = x.data_ptr(), x.shape[0], _ld(x)
        d.kvol, d.n_out = kvol, n_out
        d.weight, d.cin, d.cout = weight.data_ptr(), cin, cout
        d.bias = None if bias is None else bias.data_ptr()
        if residual is not None:
            assert residual.shape == (n_out, cout)
            d.residual, d.ld_res = residual.data_ptr(), _ld(residual)
        d.out, d.ld_out = out.data_ptr(), _ld(out)
        d.relu, d.accumulate = int(relu), int(accumulate)
        keep = _resolve_map(nbr, x, weight, d, accumulate=accumulate, stats=bn_partial, owner=w_owner)
        keep.append(_attach_workspace(d, x.device))
        partial = None
        if bn_partial:
            partial = torch.empty((max(int(lib.eprecon_conv_desc_partial_rows(ctypes.byref(d))), 1), 3, cout),
                                  dtype=torch.float32, device=x.device)
            d.bn_partial = partial.data_ptr()
        _lib.check(lib.eprecon_conv_desc_async(ctypes.byref(d), _lib.current_stream()), "eprecon_conv_desc_async")
        del keep
        return out, partial
    if nbr is not None:
        assert nbr.dtype == torch.int32 and nbr.shape[0] == kvol and nbr.is_contiguous()
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    assert out.shape == (n_out, cout)
    partial = None
    if bn_partial:
        partial = torch.empty(((n_out + 127) // 128, 3, cout), dtype=torch.float32, device=x.device)
    if residual is not None:
        assert residual.shape == (n_out, cout) and residual.dtype == torch.float32
    _lib.check(lib.eprecon_sparse_conv_fused_async(
        _lib.ptr(x), x.shape[0], _ld(x), _lib.ptr(nbr), kvol, n_out, _lib.ptr(weight), cin, cout, _lib.ptr(bias),
        _lib.ptr(residual), _ld(residual) if residual is not None else 0, _lib.ptr(out), _ld(out), int(relu),
        int(accumulate), _lib.ptr(partial), _lib.current_stream()), "eprecon_sparse_conv_fused_async")
    return out, partial


def sparse_conv_ln(x, weight, nbr, bias, ln_weight, ln_bias, ln_eps, out=None, relu=False, residual=None,
                   post_relu=False):
    """conv + bias [+ReLU] [+residual] -> row-wise LayerNorm [-> ReLU] in ONE launch (descriptor entry point):
    the spconv + LayerNorm blocks of the reference without a second pass over the tensor."""
    lib = _lib.load()
    w_owner = weight
    if weight.dim() == 2:
        weight = weight.unsqueeze(0)
    kvol, cin, cout = weight.shape
    weight = weight.contiguous()
    n_out = x.shape[0] if nbr is None else nbr.shape[1]
    assert x.shape[1] == cin and x.dtype == torch.float32 and cout <= 128
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    d = _lib.ConvDesc()
    d.x, d.n_in, d.ld_x = x.data_ptr(), x.shape[0], _ld(x)
    d.kvol, d.n_out = kvol, n_out
    keep = _resolve_map(nbr, x, weight, d, ln=True, owner=w_owner)  # noqa: F841  (alive until the launch is queued)
    d.weight, d.cin, d.cout = weight.data_ptr(), cin, cout
    d.bias = None if bias is None else bias.data_ptr()
    if residual is not None:
        assert residual.shape == (n_out, cout)
        d.residual, d.ld_res = residual.data_ptr(), _ld(residual)
    d.out, d.ld_out = out.data_ptr(), _ld(out)
    d.relu = int(relu)
    d.ln = 1
    d.ln_gamma = None if ln_weight is None else ln_weight.data_ptr()
    d.ln_beta = None if ln_bias is None else ln_bias.data_ptr()
    d.ln_eps, d.ln_post_relu = float(ln_eps), int(post_relu)
    _lib.check(lib.eprecon_conv_desc_async(ctypes.byref(d), _lib.current_stream()), "eprecon_conv_desc_async")
    return out


def affine_rows(x, scale, shift, residual=None, relu=False, out=None):
    """out = [relu]( x * scale + shift [+ residual] ): a BatchNorm whose statistics its producer finished; out may be x"""
    lib = _lib.load()
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _lib.check(lib.eprecon_affine_rows_res_async(
        _lib.ptr(x), n, c, _ld(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual),
        _ld(residual) if residual is not None else 0, int(relu), _lib.ptr(out), _ld(out), _lib.current_stream()),
        "eprecon_affine_rows_res_async")
    return out


def conv_stats(x, weight, nbr=None, in_affine=None, out=None, bias=None):
    """Convolution whose epilogue also writes the per-workgroup BatchNorm summaries of its output
    (descriptor entry point: any kernel of the family may be chosen).  in_affine = (scale, shift, relu): the
    producer's pending BatchNorm applied while gathering.  Returns (out, partial)."""
    lib = _lib.load()
    w_owner = weight
    if weight.dim() == 2:
        weight = weight.unsqueeze(0)
    kvol, cin, cout = weight.shape
    weight = weight.contiguous()
    n_out = x.shape[0] if nbr is None else nbr.shape[1]
    assert x.shape[1] == cin and x.dtype == torch.float32
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    d = _lib.ConvDesc()
    d.x, d.n_in, d.ld_x = x.data_ptr(), x.shape[0], _ld(x)
    d.kvol, d.n_out = kvol, n_out
    keep = _resolve_map(nbr, x, weight, d, stats=True, owner=w_owner)  # noqa: F841
    d.weight, d.cin, d.cout = weight.data_ptr(), cin, cout
    d.bias = None if bias is None else bias.data_ptr()
    d.out, d.ld_out = out.data_ptr(), _ld(out)
    if in_affine is not None:
        d.in_scale, d.in_shift, d.in_relu = in_affine[0].data_ptr(), in_affine[1].data_ptr(), int(in_affine[2])
    keep.append(_attach_workspace(d, x.device))
    rows = max(int(lib.eprecon_conv_desc_partial_rows(ctypes.byref(d))), 1)
    partial = torch.empty((rows, 3, cout), dtype=torch.float32, device=x.device)
    d.bn_partial = partial.data_ptr()
    if n_out > 0:
        _lib.check(lib.eprecon_conv_desc_async(ctypes.byref(d), _lib.current_stream()), "eprecon_conv_desc_async")
    return out, partial


def bn_affine(partial, gamma, beta, eps):
    """summaries -> the BatchNorm in affine form (scale, shift), to be applied by a consumer on load"""
    lib = _lib.load()
    c = partial.shape[2]
    aff = torch.empty((2, c), dtype=torch.float32, device=partial.device)
    _lib.check(lib.eprecon_batchnorm_finalize_affine_async(
        partial.data_ptr(), partial.shape[0], c, _lib.ptr(gamma), _lib.ptr(beta), float(eps), aff[0].data_ptr(),
        aff[1].data_ptr(), _lib.current_stream()), "eprecon_batchnorm_finalize_affine_async")
    return aff[0], aff[1]


def batchnorm_apply_partials(x, partial, gamma=None, beta=None, eps=1e-5, residual=None, relu=False, out=None,
                             res_affine=None):
    """second half of the train-mode BatchNorm from producer-side summaries (conv_stats);
    res_affine = (scale, shift): the residual carries a pending BatchNorm of its own, applied on load"""
    lib = _lib.load()
    n, c = x.shape
    assert partial.shape[1:] == (3, c) and partial.is_contiguous()
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    # mean / var scratch is a per-call buffer (not the shared grow-only workspace): independent branches
    # of the 2D stack run concurrently on several streams
    ws = torch.empty((lib.eprecon_batchnorm_apply_workspace_bytes(c),), dtype=torch.uint8, device=x.device)
    if res_affine is not None:
        _lib.check(lib.eprecon_batchnorm_apply_partials_res_async(
            _lib.ptr(x), n, c, _ld(x), _lib.ptr(partial), partial.shape[0], _lib.ptr(gamma), _lib.ptr(beta), float(eps),
            _lib.ptr(residual), _ld(residual), _lib.ptr(res_affine[0]), _lib.ptr(res_affine[1]), int(relu), _lib.ptr(out),
            _ld(out), _lib.ptr(ws), ws.numel(), _lib.current_stream()), "eprecon_batchnorm_apply_partials_res_async")
        return out
    _lib.check(lib.eprecon_batchnorm_apply_partials_async(
        _lib.ptr(x), n, c, _ld(x), _lib.ptr(partial), partial.shape[0], _lib.ptr(gamma), _lib.ptr(beta),
        float(eps), _lib.ptr(residual), _ld(residual) if residual is not None else 0, int(relu), _lib.ptr(out),
        _ld(out), None, None, _lib.ptr(ws), ws.numel(), _lib.current_stream()),
        "eprecon_batchnorm_apply_partials_async")
    return out


def batchnorm_train(x, gamma=None, beta=None, eps=1e-5, residual=None, relu=False, out=None):
    """train-mode BatchNorm over all rows (+ optional residual add and ReLU); out may be x"""
    lib = _lib.load()
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    ws = _lib.workspace(lib.eprecon_batchnorm_workspace_bytes(n, c), x.device)
    _lib.check(lib.eprecon_batchnorm_train_async(
        _lib.ptr(x), n, c, _ld(x), _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(residual),
        _ld(residual) if residual is not None else 0, int(relu), _lib.ptr(out), _ld(out), None, None,
        _lib.ptr(ws), ws.numel(), _lib.current_stream()), "eprecon_batchnorm_train_async")
    return out


def rowwise_layernorm(x, gamma=None, beta=None, eps=1e-5, residual=None, pre_relu=False, post_relu=False,
                      out=None):
    """y = [relu]( LN( [relu](x) [+ residual] ) * gamma + beta ) per row; out may be x"""
    lib = _lib.load()
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _lib.check(lib.eprecon_rowwise_layernorm_async(
        _lib.ptr(x), n, c, _ld(x), _lib.ptr(residual), _ld(residual) if residual is not None else 0,
        _lib.ptr(gamma), _lib.ptr(beta), float(eps), int(pre_relu), int(post_relu), _lib.ptr(out),
        _ld(out), _lib.current_stream()), "eprecon_rowwise_layernorm_async")
    return out


# ------------------------------------------------------------------------------------------------
# the per-voxel heads (Linear4xTrans) as one launch: csrc/heads.hip
# ------------------------------------------------------------------------------------------------
def _pack_mlp_weight(wt):
    """Wt f32[K][M] (in x out) -> f32[ceil(M/16)][ceil(K/16)][64][4] in the operand order of eprecon_mlp4x_async:
    block (t, c), lane 16 q + m, component i = Wt[16 c + 4 q + i][16 t + m]"""
    k, m = wt.shape
    kc, mt = (k + 15) // 16, (m + 15) // 16
    w = torch.zeros((16 * kc, 16 * mt), dtype=torch.float32, device=wt.device)
    w[:k, :m] = wt
    return w.view(kc, 4, 4, mt, 16).permute(3, 0, 1, 4, 2).contiguous().view(-1)


def _pad16(v, n=None):
    n = v.shape[0] if n is None else n
    out = torch.zeros(((n + 15) // 16 * 16,), dtype=torch.float32, device=v.device)
    out[:v.shape[0]] = v
    return out


def mlp4x_supported(channels, out_channels):
    return bool(_lib.load().eprecon_mlp4x_supported(int(channels), int(out_channels)))


def pack_mlp4x(mod):
    """the parameters of a Linear4xTrans in the kernel's operand order, ONE flat buffer (cached on the module per weight
    version; clear_packed_weights drops it) -> dict name -> tensor view"""
    params = (mod.linear1.weight, mod.linear1.bias, mod.norm1.weight, mod.norm1.bias, mod.linear2.weight, mod.linear2.bias,
              mod.norm2.weight, mod.norm2.bias, mod.linear3.weight, mod.linear3.bias)
    tag = tuple((p_._version, p_.data_ptr()) for p_ in params)
    hit = getattr(mod, "_eprecon_packed", None)
    if hit is not None and hit[0] == tag:
        return hit[1]
    with torch.no_grad():
        parts = {"w1": _pack_mlp_weight(mod.linear1.weight.t()), "b1": _pad16(mod.linear1.bias), "g1": _pad16(mod.norm1.weight),
                 "be1": _pad16(mod.norm1.bias), "w2": _pack_mlp_weight(mod.linear2.weight.t()), "b2": _pad16(mod.linear2.bias),
                 "g2": _pad16(mod.norm2.weight), "be2": _pad16(mod.norm2.bias), "w3": _pack_mlp_weight(mod.linear3.weight.t()),
                 "b3": _pad16(mod.linear3.bias)}
        flat = torch.cat([v for v in parts.values()])        # (every part is a multiple of 16 floats: 16-byte aligned views)
        views, off = {}, 0
        for name, v in parts.items():
            views[name] = flat[off:off + v.numel()]
            off += v.numel()
    mod._eprecon_packed = (tag, views)
    return views


def mlp4x(mods, x, outs=None):
    """y_h = Linear4xTrans_h(x) for the 1 or 2 modules `mods` of equal shape (TSDF and occupancy heads share their input rows):
    one launch.  x f32[n, >= C] (row pitch free), -> list of f32[n, C_out]"""
    lib = _lib.load()
    m0 = mods[0]
    c, cout = m0.linear1.in_features, m0.linear3.out_features
    n = x.shape[0]
    assert x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] >= c and 1 <= len(mods) <= 2
    d = _lib.Mlp4xDesc()
    d.x, d.ld_x, d.n = x.data_ptr(), x.stride(0), n
    d.channels, d.out_channels, d.heads, d.residual = c, cout, len(mods), int(m0.use_residual)
    d.eps1, d.eps2 = float(m0.norm1.eps), float(m0.norm2.eps)
    if outs is None:
        outs = [torch.empty((n, cout), dtype=torch.float32, device=x.device) for _ in mods]
    keep = []
    for h, (mod, y) in enumerate(zip(mods, outs)):
        assert mod.linear1.in_features == c and mod.linear3.out_features == cout
        pk = pack_mlp4x(mod)
        keep.append(pk)
        hd = d.head[h]
        for name in ("w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "w3", "b3"):
            setattr(hd, name, pk[name].data_ptr())
        hd.y, hd.ld_y = y.data_ptr(), y.stride(0)
    _lib.check(lib.eprecon_mlp4x_async(ctypes.byref(d), _lib.current_stream()), "eprecon_mlp4x_async")
    return outs
