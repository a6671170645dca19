"""Persistent sparse global map of GRU fusion: Python face of the eprecon_map_* handle
(csrc/global_map.hip), the one stateful object of the C ABI (SURVEY.md 8b "Ownership").

Mirrors the state the reference keeps in GRUFusion.global_volume[scale] / target_tsdf_volume[scale]
(models/gru_fusion.py:31-38) and the per-fragment bookkeeping of convert2dense / update_map
(:67-114,195-215).  `.C` / `.F` export copies of the rows (tests, the multi-GPU boundary exchange);
`.set(C, F)` replaces the contents.
"""
import ctypes

import torch

from . import _lib


class GruStage:
    """buffers of one queued GRU-fusion level (GlobalMap.stage_begin); read() performs the level's ONE host read, commits the
    counts to the map handles and slices the buffers to their sizes"""

    def read(self):
        from . import sparse as SP
        lib = _lib.load()
        host = _lib.read_counts(self.counts)
        n_u, kept, m1, m2 = host[:4]
        SP.check_hash_status(host[6])
        SP.check_hash_status(host[7])
        arr = (ctypes.c_int32 * 8)(*host)
        _lib.check(lib.eprecon_gru_stage_commit_async(self.map._h, self.target_map._h if self.target_map is not None else None,
                                                      ctypes.cast(arr, ctypes.c_void_p), _lib.current_stream()),
                   "eprecon_gru_stage_commit_async")
        self.n, self.n_inside, self.m1, self.m2 = n_u, self.map.size - kept, m1, m2
        for name in ("updated", "out_coords", "r_coords", "hx_v", "hx_i", "scaled1", "vox1", "inverse1", "scaled2", "vox2", "inverse2"):
            setattr(self, name, getattr(self, name)[:n_u])
        if self.tsdf_target is not None:
            self.tsdf_target = self.tsdf_target[:n_u].unsqueeze(1)
        self.uniq1, self.uniq2 = self.uniq1[:m1], self.uniq2[:m2]
        return self


class GlobalMap:
    def __init__(self, channels, device):
        lib = _lib.load()
        self.channels, self.device = int(channels), device
        h = ctypes.c_void_p()
        _lib.check(lib.eprecon_map_create(self.channels, ctypes.byref(h)), "eprecon_map_create")
        self._h = h
        # called after a stream-ordered READ of the rows has been queued (export, stamps): an owner that lets another
        # stream rewrite the map (GRUFusion's boundary exchange) re-records the event that stream waits for
        self.on_read = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().eprecon_map_destroy(h)
            except Exception:  # interpreter shutdown
                pass

    def reset(self):
        _lib.check(_lib.load().eprecon_map_reset(self._h), "eprecon_map_reset")

    @property
    def size(self):
        return int(_lib.load().eprecon_map_size(self._h))

    def export(self):
        n = self.size
        c = torch.empty((n, 3), dtype=torch.int32, device=self.device)
        f = torch.empty((n, self.channels), dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().eprecon_map_export_async(self._h, _lib.ptr(c), _lib.ptr(f), _lib.current_stream()),
                   "eprecon_map_export_async")
        if self.on_read is not None:
            self.on_read()
        return c, f

    @property
    def C(self):
        return self.export()[0]

    @property
    def F(self):
        return self.export()[1]

    @F.setter
    def F(self, feats):
        self.set(self.export()[0], feats)

    def set(self, coords, feats):
        coords = coords.to(device=self.device, dtype=torch.int32).contiguous()
        feats = feats.to(device=self.device, dtype=torch.float32).contiguous()
        assert coords.shape[0] == feats.shape[0] and feats.shape[1] == self.channels
        _lib.check(_lib.load().eprecon_map_import_async(self._h, _lib.ptr(coords), _lib.ptr(feats), coords.shape[0],
                                                        _lib.current_stream()), "eprecon_map_import_async")

    def crop_union(self, cur_coords, cur_feat, dim, interval, rel, mode=0):
        """-> (updated int32[N',3], src_cur int32[N'], src_glob int32[N'], rows of the map inside the FBV).
        Blocking (one host read for N', where the reference's torch.nonzero synchronises too)."""
        lib = _lib.load()
        n_cur = cur_feat.shape[0]
        assert cur_feat.stride(1) == 1 or n_cur == 0
        cur_coords = cur_coords.contiguous()     # raw pointers below: int32[N,4] rows
        cap = max(min(dim ** 3, n_cur + self.size), 1)
        dev = self.device
        updated = torch.empty((cap, 3), dtype=torch.int32, device=dev)
        src_cur = torch.empty(cap, dtype=torch.int32, device=dev)
        src_glob = torch.empty(cap, dtype=torch.int32, device=dev)
        rel_host = (ctypes.c_int32 * 3)(*[int(v) for v in rel])
        counts = (ctypes.c_int64 * 2)()
        _lib.count_host_read()
        _lib.check(lib.eprecon_map_crop_union(
            self._h, _lib.ptr(cur_coords), _lib.ptr(cur_feat), n_cur, cur_feat.stride(0) if n_cur else self.channels,
            dim, interval, int(mode), ctypes.cast(rel_host, ctypes.c_void_p), _lib.ptr(updated), _lib.ptr(src_cur),
            _lib.ptr(src_glob), ctypes.cast(counts, ctypes.c_void_p), _lib.current_stream()), "eprecon_map_crop_union")
        n = int(counts[0])
        return updated[:n], src_cur[:n], src_glob[:n], int(counts[1])

    def stage_begin(self, target_map, cur_coords, cur_feat, dim, interval, rel, tsdf_gt, occ_gt, origin, w2ac, voxel_size,
                    resolution, ch_voxel, batch_index=0, mode=0):
        """One GRU-fusion level queued as ONE call with device-side counts (eprecon_gru_stage_begin_async): crop + union,
        the [h | x] rows of both ConvGRUs, the ground-truth twin's targets, the fragment's points in the aligned-camera frame
        and the coordinate side of the two shared voxelisations.  Nothing waits for the device; -> GruStage (read() is the one
        host read of the level's bookkeeping)."""
        from . import sparse as SP
        lib = _lib.load()
        dev = self.device
        n_cur = cur_feat.shape[0]
        assert cur_feat.stride(1) == 1 or n_cur == 0
        cur_coords = cur_coords.contiguous()
        cap = int(lib.eprecon_gru_stage_capacity(self._h, n_cur, dim))
        c, chv = self.channels, int(ch_voxel)
        st = GruStage()
        st.map, st.target_map, st.cap = self, target_map, cap
        i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        st.updated, st.out_coords, st.r_coords = i32(cap, 3), i32(cap, 4), f32(cap, 4)
        st.hx_v, st.hx_i = f32(cap, 2 * chv), f32(cap, 2 * (c - chv))
        st.tsdf_target = f32(cap) if target_map is not None else None
        st.scaled1, st.vox1, st.inverse1, st.uniq1, st.grid1 = f32(cap, 4), i32(cap, 4), i32(cap), i32(cap, 4), SP.HashGrid(cap, dev)
        st.scaled2, st.vox2, st.inverse2, st.uniq2, st.grid2 = f32(cap, 4), i32(cap, 4), i32(cap), i32(cap, 4), SP.HashGrid(cap, dev)
        st.counts = i32(8)
        ws = _lib.workspace(lib.eprecon_gru_stage_workspace_bytes(cap), dev)
        d = _lib.GruStageDesc()
        d.map, d.target_map = self._h, (target_map._h if target_map is not None else None)
        d.cur_coords, d.cur_feat, d.n_cur, d.ld_cur = _lib.ptr(cur_coords), _lib.ptr(cur_feat), n_cur, (cur_feat.stride(0) if n_cur else c)
        d.dim, d.interval, d.activity_mode = int(dim), int(interval), int(mode)
        d.rel[0], d.rel[1], d.rel[2] = (int(v) for v in rel)
        keep = [cur_coords, cur_feat, ws]
        if target_map is not None:
            tg = tsdf_gt.to(torch.float32).contiguous()
            og = occ_gt.contiguous().view(torch.uint8) if occ_gt.dtype == torch.bool else occ_gt.to(torch.uint8).contiguous()
            d.tsdf_gt, d.occ_gt = _lib.ptr(tg), _lib.ptr(og)
            keep += [tg, og]
        origin = origin.detach().to(torch.float32).reshape(3).contiguous()
        w2ac = w2ac.detach().to(torch.float32).reshape(16).contiguous()
        keep += [origin, w2ac]
        d.origin, d.w2ac = _lib.ptr(origin), _lib.ptr(w2ac)
        d.voxel_size, d.resolution, d.ch_voxel, d.batch_index, d.capacity = float(voxel_size), float(resolution), chv, int(batch_index), cap
        d.updated, d.out_coords, d.r_coords = _lib.ptr(st.updated), _lib.ptr(st.out_coords), _lib.ptr(st.r_coords)
        d.hx_voxel, d.hx_image, d.tsdf_target = _lib.ptr(st.hx_v), _lib.ptr(st.hx_i), _lib.ptr(st.tsdf_target)
        d.scaled1, d.vox1, d.inverse1, d.uniq1, d.table1 = (_lib.ptr(t) for t in (st.scaled1, st.vox1, st.inverse1, st.uniq1, st.grid1.mem))
        d.scaled2, d.vox2, d.inverse2, d.uniq2, d.table2 = (_lib.ptr(t) for t in (st.scaled2, st.vox2, st.inverse2, st.uniq2, st.grid2.mem))
        d.table_capacity = st.grid1.capacity
        d.counts = _lib.ptr(st.counts)
        d.workspace, d.workspace_bytes = _lib.ptr(ws), ws.numel()
        _lib.check(lib.eprecon_gru_stage_begin_async(ctypes.byref(d), _lib.current_stream()), "eprecon_gru_stage_begin_async")
        st._keep = keep
        return st

    def gather(self, src_glob, col0, channels, out, fill=0.0):
        """out[i] = map.F[src_glob[i], col0:col0+channels] (fill where src_glob[i] < 0); out may be a column slice"""
        _lib.check(_lib.load().eprecon_map_gather_async(self._h, _lib.ptr(src_glob), src_glob.shape[0], int(col0),
                                                        int(channels), float(fill), _lib.ptr(out), out.stride(0),
                                                        _lib.current_stream()), "eprecon_map_gather_async")
        return out

    def update(self, updated, values):
        """update_map (models/gru_fusion.py:195-215) after crop_union: rows inside the FBV are replaced by
        (updated + relative origin, values)"""
        _lib.check(_lib.load().eprecon_map_update_async(self._h, _lib.ptr(updated), updated.shape[0], _lib.ptr(values),
                                                        values.stride(0), _lib.current_stream()),
                   "eprecon_map_update_async")

    # ---- multi-GPU boundary exchange (eprecon_amd/distributed.py) ----
    def set_fragment(self, fragment_index):
        """global index of the fragment whose fusion the next update() appends (-1: rows carry no origin)"""
        _lib.check(_lib.load().eprecon_map_set_fragment(self._h, int(fragment_index)), "eprecon_map_set_fragment")

    def stamps(self):
        """int32[size]: 0 unknown, +(fragment + 1) fused by this rank, -(fragment + 1) received"""
        out = torch.empty(self.size, dtype=torch.int32, device=self.device)
        _lib.check(_lib.load().eprecon_map_stamps_async(self._h, _lib.ptr(out), None, 0, 0, _lib.current_stream()),
                   "eprecon_map_stamps_async")
        if self.on_read is not None:
            self.on_read()
        return out

    def set_stamps(self, stamps=None, fill=None):
        src = None if stamps is None else stamps.to(device=self.device, dtype=torch.int32).contiguous()
        assert src is None or src.shape[0] == self.size
        _lib.check(_lib.load().eprecon_map_stamps_async(self._h, None, _lib.ptr(src), int(fill is not None), int(fill or 0),
                                                        _lib.current_stream()), "eprecon_map_stamps_async")

    def select_boundary(self, boxes_lo, own_box, dim, count_out):
        """boxes_lo int32[n_boxes,3] on the device; count_out: a device int32 element that receives the number of rows"""
        _lib.check(_lib.load().eprecon_map_select_boundary_async(self._h, _lib.ptr(boxes_lo), boxes_lo.shape[0], int(own_box), int(dim),
                                                                 _lib.ptr(count_out), _lib.current_stream()),
                   "eprecon_map_select_boundary_async")

    def pack_boundary(self, payload, n_rows):
        """payload: f32 buffer with room for n_rows x (4 + channels)"""
        _lib.check(_lib.load().eprecon_map_pack_boundary_async(self._h, _lib.ptr(payload), int(n_rows), _lib.current_stream()),
                   "eprecon_map_pack_boundary_async")

    def merge_boundary(self, payload, n_rows, box_lo, dim):
        """-> number of rows appended (blocking)"""
        lo = (ctypes.c_int32 * 3)(*[int(v) for v in box_lo])
        added = ctypes.c_int64(0)
        _lib.count_host_read()
        _lib.check(_lib.load().eprecon_map_merge_boundary(self._h, _lib.ptr(payload), int(n_rows), ctypes.cast(lo, ctypes.c_void_p),
                                                          int(dim), ctypes.cast(ctypes.byref(added), ctypes.c_void_p),
                                                          _lib.current_stream()), "eprecon_map_merge_boundary")
        return int(added.value)

    def target_fuse(self, tsdf_gt, occ_gt, dim, rel, updated):
        """ground-truth twin (1 channel): -> tsdf_target f32[N',1] at the union voxels; the map is updated"""
        lib = _lib.load()
        tsdf_gt = tsdf_gt.to(torch.float32).contiguous()
        occ_u8 = occ_gt.contiguous().view(torch.uint8) if occ_gt.dtype == torch.bool else occ_gt.to(torch.uint8).contiguous()
        out = torch.empty((updated.shape[0], 1), dtype=torch.float32, device=self.device)
        rel_host = (ctypes.c_int32 * 3)(*[int(v) for v in rel])
        _lib.count_host_read()
        _lib.check(lib.eprecon_map_target_fuse(self._h, _lib.ptr(tsdf_gt), _lib.ptr(occ_u8), dim,
                                               ctypes.cast(rel_host, ctypes.c_void_p), _lib.ptr(updated), updated.shape[0],
                                               _lib.ptr(out), _lib.current_stream()), "eprecon_map_target_fuse")
        return out
