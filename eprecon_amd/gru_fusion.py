"""GRU fusion of fragment features into a persistent sparse global map — mirror of
models/gru_fusion.py (GRUFusion, feature mode :259-394) on libeprecon_hip.so.

Per scale the module keeps the global map as {C int32[M,3] (scene-grid voxel units of that
scale), F f32[M,C]} plus its ground-truth TSDF twin (the reference's test path reads the targets,
models/neucon_network.py:488).  One fragment step (per batch element):

  relative origin -> union of current voxels and the in-FBV part of the map, raster order
  (the map is a C-ABI handle, eprecon_amd/global_map.py + csrc/global_map.hip: index volumes + scan, no dense
  feature volume, no per-fragment re-allocation) -> gather current / global
  rows -> aligned-camera coordinates -> ConvGRU on the voxel channels and ConvGRU on the image
  channels (eprecon_amd.modules.ConvGRU) -> write the fused rows back into the map.

`direct_substitute=True` (scene-level TSDF substitution + instance association, a15) dispatches to
eprecon_amd/scene_fusion.py.
"""
import torch
import torch.nn as nn

from . import _lib
from .global_map import GlobalMap
from .modules import ConvGRU
from .tensor import PointTensor
from .torchsparse_utils import (aligned_camera_coords, convgru_resolution, prepare_convgru_voxelizations,
                                register_voxelization, register_voxelization_pair)


def fbv_union(cur_coords, cur_feat, glob_coords, glob_feat, dim, interval, rel, mode=0):
    """-> (updated int32[N',3], src_cur int32[N'], src_glob int32[N'], glob_valid bool[M]).
    One host sync for N' (the reference's torch.nonzero syncs at the same point)."""
    lib = _lib.load()
    dev = cur_feat.device
    n_cur, c = cur_feat.shape
    n_glob = glob_coords.shape[0]
    cells = dim ** 3
    cap = min(cells, n_cur + n_glob)
    updated = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    src_cur = torch.empty(cap, dtype=torch.int32, device=dev)
    src_glob = torch.empty(cap, dtype=torch.int32, device=dev)
    glob_valid = torch.zeros(max(n_glob, 1), dtype=torch.uint8, device=dev)
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _lib.workspace(lib.eprecon_fbv_union_workspace_bytes(dim), dev)
    import ctypes
    rel_host = (ctypes.c_int32 * 3)(*[int(v) for v in rel])
    gf = glob_feat if n_glob > 0 else None
    _lib.check(lib.eprecon_fbv_union_async(
        _lib.ptr(cur_coords), _lib.ptr(cur_feat), n_cur, cur_feat.stride(0) if n_cur else c,
        _lib.ptr(glob_coords) if n_glob else None, _lib.ptr(gf), n_glob, glob_feat.stride(0) if n_glob else c,
        c, dim, interval, int(mode), ctypes.cast(rel_host, ctypes.c_void_p), _lib.ptr(updated), _lib.ptr(src_cur),
        _lib.ptr(src_glob), _lib.ptr(glob_valid), _lib.ptr(n_out), _lib.ptr(ws), ws.numel(),
        _lib.current_stream()), "eprecon_fbv_union_async")
    n = int(n_out.item())
    return updated[:n], src_cur[:n], src_glob[:n], glob_valid[:n_glob].bool()


def gather_rows(feat, src, channels, fill=0.0, out=None):
    """out[i] = feat[src[i]] (or `fill` where src[i] < 0); feat / out may be column slices of wider buffers"""
    lib = _lib.load()
    n = src.shape[0]
    if out is None:
        out = torch.empty((n, channels), dtype=torch.float32, device=src.device)
    has = feat is not None and feat.shape[0] > 0
    if not has:
        return out.fill_(fill)
    _lib.check(lib.eprecon_gather_rows_async(_lib.ptr(feat), feat.stride(0), _lib.ptr(src), n, channels,
                                             float(fill), _lib.ptr(out), out.stride(0), _lib.current_stream()),
               "eprecon_gather_rows_async")
    return out


class GRUFusion(nn.Module):
    def __init__(self, cfg, ch_in=None, direct_substitute=False, trianing=True, ch_voxel=None):
        super().__init__()
        self.cfg = cfg
        self.direct_substitude = bool(direct_substitute)
        if direct_substitute:
            # scene-level TSDF substitution + instance association (NeuralRecon.fuse_to_global)
            from .scene_fusion import SceneFusion
            self._scene = SceneFusion(cfg)
            return
        self.ch_in = list(ch_in)
        self.feat_init = 0
        self.ch_voxel = list(ch_voxel)
        self.ch_img = [a - b for a, b in zip(ch_in, ch_voxel)]
        self.n_scales = len(cfg.THRESHOLDS) - 1
        self.scene_name = [None, None, None]
        self.global_origin = [None, None, None]
        self.global_volume = [None, None, None]
        self.target_tsdf_volume = [None, None, None]
        self.coords_dtype = torch.int32  # the reference returns int64; int32 is this package's coordinate type
        self.fusion_nets_voxel = nn.ModuleList()
        self.fusion_nets_img = nn.ModuleList()
        for i, ch in enumerate(self.ch_voxel):
            self.fusion_nets_voxel.append(ConvGRU(hidden_dim=ch, input_dim=ch, pres=1,
                                                  vres=cfg.VOXEL_SIZE * 2 ** (self.n_scales - i)))
        for i, ch in enumerate(self.ch_img):
            self.fusion_nets_img.append(ConvGRU(hidden_dim=ch, input_dim=ch, pres=1,
                                                vres=cfg.VOXEL_SIZE * 2 ** (self.n_scales - i)))
        self._identity_fusion = False  # tests: skip the ConvGRUs (pins the bookkeeping alone)
        self.two_streams = __import__("os").environ.get("EPRECON_GRU_STREAMS", "1") == "1"
        # EPRECON_GRU_STAGE=0: the level's bookkeeping as separate calls with a host read each (crop_union, target_fuse, the
        # two unique-voxel counts) instead of ONE queued stage call + one read (GlobalMap.stage_begin)
        self.stage_call = __import__("os").environ.get("EPRECON_GRU_STAGE", "1") == "1"
        self._side = None
        self._xchg = None              # multi-GPU: distributed.BoundaryExchange (stamps of the map voxels)
        self._n_exchanged = self._cur_fragment = 0
        self._xchg_stream = None       # the exchange's own stream: its one host read must not drain the main stream
        self._map_ready = None         # event behind the last map update (what the exchange has to wait for)

    def reset(self, i, device=None):
        """models/gru_fusion.py:59-65: empty maps (the handles keep their device memory)"""
        device = device or torch.device("cuda")
        if self.global_volume[i] is None or self.global_volume[i].device != device:
            self.global_volume[i] = GlobalMap(self.ch_in[i], device)
            self.target_tsdf_volume[i] = GlobalMap(1, device)
            self.global_volume[i].on_read = self._note_map_read
        self.global_volume[i].reset()
        self.target_tsdf_volume[i].reset()
        if self._xchg is not None:
            self._xchg.stamps[i] = self._xchg.new_stamps(i)

    def _note_map_read(self):
        """a read of the map rows was queued on the current stream (export / stamps): the exchange stream, which rewrites
        rows in place, must also wait for it"""
        if self._xchg is not None and self._xchg_stream is not None and torch.cuda.is_available():
            self._map_ready = torch.cuda.current_stream().record_event()

    def _begin_fragment(self, scale, inputs, i, dev):
        """scene bookkeeping of models/gru_fusion.py:280-293 -> relative origin (LongTensor[3], host)"""
        scene = inputs["scene"][i]
        vol_origin, partial = self._host_origins(inputs)
        if self.scene_name[scale] is None or scene != self.scene_name[scale]:
            self.scene_name[scale] = scene
            self.reset(scale, dev)
            self.global_origin[scale] = vol_origin[i].clone()
        interval = 2 ** (self.cfg.N_LAYER - scale - 1)
        # (origin - global_origin) / voxel_size in fp32, truncated toward zero (:292-293)
        return ((partial[i] - self.global_origin[scale]) / (self.cfg.VOXEL_SIZE * interval)).long()

    def _host_origins(self, inputs):
        """host copies of inputs['vol_origin'] / ['vol_origin_partial'] (f32[B,3]), fetched once per fragment
        instead of once per scale (each .cpu() is a device synchronisation).  A caller that already holds the origins on
        the host (the data loader builds them there: datasets/transforms.py:250-260) passes them as
        inputs['vol_origin_host'] / ['vol_origin_partial_host'] and no device read happens at all.
        The device read is issued on the CURRENT stream: callers that switch streams (exchange_boundaries) call this on
        the stream the inputs were produced on first, and hit the cache afterwards."""
        vo, vp = inputs["vol_origin"], inputs["vol_origin_partial"]
        key = (vo.data_ptr(), vo._version, vp.data_ptr(), vp._version)
        hit = getattr(self, "_origin_cache", None)
        if hit is None or hit[0] != key or hit[1] is not vo or hit[2] is not vp:
            ho, hp = inputs.get("vol_origin_host"), inputs.get("vol_origin_partial_host")
            if ho is not None and hp is not None:
                both = (torch.as_tensor(ho).detach().float().reshape(-1, 3).cpu(),
                        torch.as_tensor(hp).detach().float().reshape(-1, 3).cpu())
            else:
                _lib.count_host_read()
                both = torch.stack([vo.detach().float().reshape(-1, 3), vp.detach().float().reshape(-1, 3)]).cpu()
            hit = (key, vo, vp, both[0], both[1])
            self._origin_cache = hit
        return hit[3], hit[4]

    def exchange_boundaries(self, inputs, i=0):
        """Multi-GPU schedule (eprecon_amd/distributed.py): before this fragment is fused, pull in the map voxels
        other ranks fused inside this fragment's bounding volume — all three scales in one exchange (three
        collectives, one host read).  Collective: every rank calls it once per fragment, outside any
        data-dependent control flow."""
        import torch.distributed as dist
        from . import distributed as D
        dev = inputs["vol_origin_partial"].device
        n = self.cfg.N_LAYER
        if self._xchg is None:
            self._xchg = D.BoundaryExchange(n, dev)
        if dev.type == "cuda" and __import__("os").environ.get("EPRECON_XCHG_STREAM", "1") == "1":
            # The exchange reads the maps as the PREVIOUS fragment's last update left them and needs one host read (the
            # counts).  On the main stream that read would wait for everything queued there — the host runs milliseconds
            # ahead of the GPU — and end the run-ahead at the start of every fragment.  On its own stream it waits only for
            # the event behind the last map update; the main stream joins it afterwards (the merges change the maps).
            if self._xchg_stream is None:
                self._xchg_stream = _lib.side_stream(dev, _lib.SIDE_EXCHANGE)
            main = torch.cuda.current_stream(dev)
            # The origins are inputs of THIS fragment: produced / uploaded on the main stream after `_map_ready` was recorded,
            # so the side stream is not ordered behind them.  Their host copy is therefore taken here, on the main stream
            # (no device read at all when the caller passes the host-side origins), and the side stream only hits the cache.
            self._host_origins(inputs)
            if self._map_ready is not None:
                self._xchg_stream.wait_event(self._map_ready)
            else:
                self._xchg_stream.wait_stream(main)
            with torch.cuda.stream(self._xchg_stream):
                self._exchange_on_current_stream(inputs, i, dev, n, dist)
            main.wait_stream(self._xchg_stream)
            return
        self._exchange_on_current_stream(inputs, i, dev, n, dist)

    def _exchange_on_current_stream(self, inputs, i, dev, n, dist):
        rels, dims = [], []
        for scale in range(n):
            rels.append(self._begin_fragment(scale, inputs, i, dev).tolist())
            dims.append(self.cfg.N_VOX[0] // 2 ** (n - scale - 1))
        # global index of the fragment this rank fuses next: ranks take the fragments of a scene round-robin
        self._cur_fragment = self._n_exchanged * dist.get_world_size() + dist.get_rank()
        self._n_exchanged += 1
        if dev.type == "cuda":
            # selection / packing / merge as kernels on the map handles; the stamps are a column of the map
            self._xchg.exchange_handles([self.global_volume[scale] for scale in range(n)], rels, dims)
            for scale in range(n):
                self.global_volume[scale].set_fragment(self._cur_fragment)
            return
        maps = [self.global_volume[scale].export() for scale in range(n)]
        for scale, (c, f) in enumerate(self._xchg.exchange(maps, rels, dims)):
            if c.shape[0] != maps[scale][0].shape[0] or f is not maps[scale][1]:
                self.global_volume[scale].set(c, f)

    def _fuse_staged(self, scale, i, cur_c, cur_f, dim, interval, rel_l, origin, inputs, dev):
        """one batch element of one level on the stage call (inference on the GPU): -> (coords, fused values, tsdf_target,
        occ_target).  One host read for the whole bookkeeping; the ConvGRUs find their voxelisations in the cache."""
        cfg = self.cfg
        gmap = self.global_volume[scale]
        chv, cin = self.ch_voxel[scale], self.ch_in[scale]
        chi = cin - chv
        gv, gi = self.fusion_nets_voxel[scale], self.fusion_nets_img[scale]
        res = convgru_resolution(gv.convz.pres, gv.convz.vres)
        tmap = tsdf_gt = occ_gt = None
        if "occ_list" in inputs:
            lvl = cfg.N_LAYER - scale - 1
            tmap, tsdf_gt, occ_gt = self.target_tsdf_volume[scale], inputs["tsdf_list"][lvl][i], inputs["occ_list"][lvl][i]
        st = gmap.stage_begin(tmap, cur_c, cur_f, dim, interval, rel_l, tsdf_gt, occ_gt, origin,
                              inputs["world_to_aligned_camera"][i], cfg.VOXEL_SIZE, res, chv, batch_index=i).read()
        n_u = st.n
        r_coords = st.r_coords
        if self._identity_fusion:      # tests: the bookkeeping alone
            e1 = register_voxelization(r_coords, res, st.scaled1, st.vox1, st.inverse1, st.uniq1, st.grid1)
            register_voxelization(e1.scaled, res, st.scaled2, st.vox2, st.inverse2, st.uniq2, st.grid2)
        else:
            # point lists, kernel maps, corner tables (and the reference's hash order / stale indices) of both voxelisations:
            # one library call; prepare_convgru_voxelizations below then finds everything in place
            register_voxelization_pair(r_coords, res, (st.scaled1, st.vox1, st.inverse1, st.uniq1, st.grid1),
                                       (st.scaled2, st.vox2, st.inverse2, st.uniq2, st.grid2))
        values = torch.empty((n_u, cin), dtype=torch.float32, device=dev)
        hx_v, hx_i = st.hx_v, st.hx_i
        if self._identity_fusion:          # tests: the bookkeeping alone (the fragment's rows pass through)
            values[:, :chv], values[:, chv:] = hx_v[:, chv:], hx_i[:, chi:]
        else:
            prepare_convgru_voxelizations(r_coords, gv.convz.pres, gv.convz.vres)   # kernel maps, point lists, corner tables
        if self._identity_fusion:
            pass
        elif self.two_streams:
            main = torch.cuda.current_stream(dev)
            if self._side is None:
                self._side = _lib.side_stream(dev, _lib.SIDE_PAIR)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                gi(PointTensor(hx_i[:, :chi], r_coords), PointTensor(hx_i[:, chi:], r_coords), out=values[:, chv:])
            gv(PointTensor(hx_v[:, :chv], r_coords), PointTensor(hx_v[:, chv:], r_coords), out=values[:, :chv])
            main.wait_stream(self._side)
        else:
            gv(PointTensor(hx_v[:, :chv], r_coords), PointTensor(hx_v[:, chv:], r_coords), out=values[:, :chv])
            gi(PointTensor(hx_i[:, :chi], r_coords), PointTensor(hx_i[:, chi:], r_coords), out=values[:, chv:])
        gmap.update(st.updated, values)                                          # update_map (:195-215)
        if self._xchg is not None:
            self._map_ready = torch.cuda.current_stream(dev).record_event()
        tsdf_target = st.tsdf_target
        occ_target = tsdf_target.abs() < 1 if tsdf_target is not None else None
        return st.out_coords, values, tsdf_target, occ_target

    def forward(self, coords, values_in, inputs, scale=2, outputs=None, save_mesh=False, panoptic_infos=None):
        """coords int[N,4] (b,x,y,z) finest units, values_in f32[N,C] ->
        (coords[N',4] raster order per batch element, fused f32[N',C], tsdf_target f32[N',1] | None,
        occ_target bool[N',1] | None)   (models/gru_fusion.py:259-394)"""
        if self.direct_substitude:
            return self._scene.forward(coords, values_in, inputs, scale, outputs, save_mesh, panoptic_infos)
        cfg = self.cfg
        batch_size = len(inputs["fragment"])
        interval = 2 ** (cfg.N_LAYER - scale - 1)
        dim = cfg.N_VOX[0] // interval
        voxel_size = cfg.VOXEL_SIZE * interval
        dev = values_in.device
        coords = coords if coords.dtype == torch.int32 else coords.to(torch.int32)
        chv = self.ch_voxel[scale]
        out_c, out_v, out_t, out_o = [], [], [], []
        if batch_size == 1:
            bounds = [0, coords.shape[0]]
        else:  # rows are grouped by ascending batch index (models/neucon_network.py:246-251): one host read
            bounds = [0] + torch.cumsum(torch.bincount(coords[:, 0].long(), minlength=batch_size), 0).tolist()
        for i in range(batch_size):
            rel = self._begin_fragment(scale, inputs, i, dev)
            origin = inputs["vol_origin_partial"][i]
            lo, hi = bounds[i], bounds[i + 1]
            if hi == lo:
                continue
            cur_c, cur_f = coords[lo:hi], values_in[lo:hi]
            gmap = self.global_volume[scale]
            rel_l = rel.tolist()
            if self.stage_call and dev.type == "cuda" and not torch.is_grad_enabled() and cur_f.stride(1) == 1:
                res = self._fuse_staged(scale, i, cur_c, cur_f, dim, interval, rel_l, origin, inputs, dev)
                for part, lst in zip(res, (out_c, out_v, out_t, out_o)):
                    if part is not None:
                        lst.append(part)
                continue
            updated, src_cur, src_glob, _ = gmap.crop_union(cur_c, cur_f, dim, interval, rel_l)
            n_u, cin = updated.shape[0], self.ch_in[scale]
            chi = cin - chv
            recording = torch.is_grad_enabled() and values_in.requires_grad
            if recording:
                # training: the map rows are state (the reference detaches the map at the start of every forward,
                # models/gru_fusion.py:260-263); the fragment's rows keep their graph through an index gather
                h_all = gmap.gather(src_glob, 0, cin, torch.empty((n_u, cin), dtype=torch.float32, device=dev))
                pad = torch.cat([cur_f, cur_f.new_zeros((1, cin))])
                x_all = pad[torch.where(src_cur >= 0, src_cur, torch.full_like(src_cur, cur_f.shape[0])).long()]
            else:
                # current / global rows gathered straight into the [h | x] buffers of the two ConvGRUs (voxel
                # channels and image channels): h = the map's row, x = the fragment's row; no concat / slice copies
                hx_v = torch.empty((n_u, 2 * chv), dtype=torch.float32, device=dev)
                hx_i = torch.empty((n_u, 2 * chi), dtype=torch.float32, device=dev)
                gmap.gather(src_glob, 0, chv, hx_v[:, :chv])
                gmap.gather(src_glob, chv, chi, hx_i[:, :chi])
                gather_rows(cur_f[:, :chv], src_cur, chv, out=hx_v[:, chv:])
                gather_rows(cur_f[:, chv:], src_cur, chi, out=hx_i[:, chi:])

            tsdf_target = occ_target = None
            if "occ_list" in inputs:
                lvl = cfg.N_LAYER - scale - 1
                tsdf_target = self.target_tsdf_volume[scale].target_fuse(
                    inputs["tsdf_list"][lvl][i], inputs["occ_list"][lvl][i], dim, rel_l, updated)
                occ_target = tsdf_target.abs() < 1

            values = None if recording else torch.empty((n_u, cin), dtype=torch.float32, device=dev)
            pts_c = None
            if recording and self._identity_fusion:
                values = x_all
            elif recording:
                pts_c = torch.cat([torch.zeros_like(updated[:, :1]), updated * interval], dim=1)
                r_coords = aligned_camera_coords(pts_c, origin.reshape(1, 3), cfg.VOXEL_SIZE,
                                                 inputs["world_to_aligned_camera"][i].reshape(1, 4, 4))
                gv, gi = self.fusion_nets_voxel[scale], self.fusion_nets_img[scale]
                vv = gv(PointTensor(h_all[:, :chv], r_coords), PointTensor(x_all[:, :chv], r_coords))
                vi = gi(PointTensor(h_all[:, chv:], r_coords), PointTensor(x_all[:, chv:], r_coords))
                values = torch.cat([vv, vi], dim=1)
            elif not self._identity_fusion:
                pts_c = torch.cat([torch.zeros_like(updated[:, :1]), updated * interval], dim=1)
                r_coords = aligned_camera_coords(pts_c, origin.reshape(1, 3), cfg.VOXEL_SIZE,
                                                 inputs["world_to_aligned_camera"][i].reshape(1, 4, 4))
                # both cells see the same points: their six SConv3d share two voxelisations (torchsparse_utils)
                gv, gi = self.fusion_nets_voxel[scale], self.fusion_nets_img[scale]
                if self.two_streams and dev.type == "cuda" and not torch.is_grad_enabled():
                    # the shared voxelisations are built first; the two cells (independent channel groups) then run
                    # on two HIP streams: each is a chain of small kernels that leaves most CUs idle on its own
                    prepare_convgru_voxelizations(r_coords, gv.convz.pres, gv.convz.vres)
                    main = torch.cuda.current_stream(dev)
                    if self._side is None:
                        self._side = _lib.side_stream(dev, _lib.SIDE_PAIR)
                    self._side.wait_stream(main)
                    with torch.cuda.stream(self._side):
                        gi(PointTensor(hx_i[:, :chi], r_coords), PointTensor(hx_i[:, chi:], r_coords), out=values[:, chv:])
                    gv(PointTensor(hx_v[:, :chv], r_coords), PointTensor(hx_v[:, chv:], r_coords), out=values[:, :chv])
                    main.wait_stream(self._side)
                else:
                    gv(PointTensor(hx_v[:, :chv], r_coords), PointTensor(hx_v[:, chv:], r_coords), out=values[:, :chv])
                    gi(PointTensor(hx_i[:, :chi], r_coords), PointTensor(hx_i[:, chi:], r_coords), out=values[:, chv:])
            else:
                values[:, :chv], values[:, chv:] = hx_v[:, chv:], hx_i[:, chi:]

            gmap.update(updated, values.detach().contiguous() if recording else values)    # update_map (:195-215)
            if self._xchg is not None and dev.type == "cuda":
                self._map_ready = torch.cuda.current_stream(dev).record_event()
            if self._xchg is not None and dev.type != "cuda":   # (reference path; on the GPU update() stamps the rows itself)
                rel_t = torch.tensor(rel_l, dtype=torch.int32, device=dev)
                self._xchg.mark_fused(scale, updated + rel_t, self._cur_fragment)

            # (batch element 0: the rows the aligned-camera transform was given are these coordinates already)
            out_c.append(pts_c if (i == 0 and pts_c is not None) else
                         torch.cat([torch.full_like(updated[:, :1], i), updated * interval], dim=1))
            out_v.append(values)
            if tsdf_target is not None:
                out_t.append(tsdf_target)
                out_o.append(occ_target)
        if not out_c:
            return None, None, None, None
        cat1 = lambda parts: parts[0] if len(parts) == 1 else torch.cat(parts)     # (one batch element: no copy)
        coords_all = cat1(out_c).to(self.coords_dtype)
        return (coords_all, cat1(out_v), cat1(out_t) if out_t else None, cat1(out_o) if out_o else None)
