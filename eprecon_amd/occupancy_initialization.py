"""Occupancy initialisation ("depth prior") — mirror of models/occupancy_initialization.py:11-182.

Per batch element: fuse the three pyramid levels of the 9 views into 32-channel 1/8-resolution maps
(dense 2D convolutions, PyTorch-ROCm), back-project them onto the dense 48^3 grid and take the
per-voxel variance over the visible views (HIP, csrc/back_project.hip), then run the submanifold
stack BN -> sparse ELAN -> 3 x (SubM3 + ReLU + residual + LN) -> SubM3(32->1) -> BN on the voxels
seen by >= min_view views (HIP: hash-grid kernel map built once, MFMA gather-GEMM convolutions,
fused normalisation epilogues).  Returns [occupancy logit f32[N_valid,1], coords[N_valid,4],
count f32[N]] or None when a batch element has fewer than 1000 valid voxels (:107-108).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import back_project as BP
# the reference defines these two in THIS module (models/occupancy_initialization.py:185,264; imported from it by
# models/neucon_network.py:20): re-exported so that `from models.occupancy_initialization import Occupancy_Initialization,
# Back_Project, get_img_feats` needs nothing but the package name swapped
from .back_project import Back_Project, get_img_feats  # noqa: F401
from . import dense2d as D2
from . import sparse as SP
from .config import INIT_MIN_VALID
from .modules import (Conv2d_Block, Conv2d_Residual_Block, Fusion_Block, SparseSubMConv3d, Spares3dELAN,
                      TrainBatchNorm1d, _RowLayerNorm, upsample2x_bilinear)


class Occupancy_Initialization(nn.Module):
    def __init__(self, ch_initialization_all, ch_initialization_down, n_views):
        super().__init__()
        ch_all = sum(ch_initialization_all[:3])
        d = ch_initialization_down
        self.self_fusion_1x = Fusion_Block(ch_initialization_all[0])
        self.self_fusion_2x = Fusion_Block(ch_initialization_all[1])
        self.self_fusion_4x = Fusion_Block(ch_initialization_all[2])
        self.pool4x = nn.AvgPool2d(2)
        self.fusion_down = Conv2d_Block(ch_all, d, 1)
        self.post_fusion_1 = Conv2d_Residual_Block(d, 3)
        self.post_fusion_2 = Conv2d_Residual_Block(d, 3)
        self.post_fusion_3 = Conv2d_Residual_Block(d, 3)
        self.post_fusion_4 = Conv2d_Residual_Block(d, 3)

        self.similary_1 = Spares3dELAN(d)
        self.norm0 = TrainBatchNorm1d(d)
        self.subm1 = SparseSubMConv3d(d, d, 3)
        self.norm1 = _RowLayerNorm(d)
        self.subm2 = SparseSubMConv3d(d, d, 3)
        self.norm2 = _RowLayerNorm(d)
        self.subm3 = SparseSubMConv3d(d, d, 3)
        self.norm3 = _RowLayerNorm(d)
        self.subm4 = SparseSubMConv3d(d, 1, 3)
        self.norm4 = TrainBatchNorm1d(1)
        # The 2D fusion stack is ~60 small MIOpen / elementwise launches on static shapes: under
        # torch.no_grad() it is captured once into a HIP graph and replayed (launch-bound otherwise).
        self.use_hip_graph = os.environ.get("EPRECON_NO_GRAPH", "0") != "1"
        self._graphs = {}
        # EPRECON_MIOPEN_CONV2D=1 keeps the 2D convolutions on PyTorch-ROCm / MIOpen (A/B switch)
        self.use_hip_conv = os.environ.get("EPRECON_MIOPEN_CONV2D", "0") != "1"
        self._channels_last = False
        self._side_streams = None
        self._bn_arena = None       # accumulator blocks of the 2D stack's BatchNorms (dense2d.BnArena), one per module

    def feat_fusion_pre(self, feats_1x, feats_2x, feats_4x):
        """[V,80,H/16,W/16], [V,40,H/8,W/8], [V,24,H/4,W/4] -> [V,32,H/8,W/8]  (:41-58)"""
        if feats_1x.is_cuda and not torch.is_grad_enabled() and self.use_hip_conv:
            return self._feat_fusion_rows(feats_1x, feats_2x, feats_4x)
        if feats_1x.is_cuda and not torch.is_grad_enabled():
            # MIOpen NHWC convolutions + HIP BatchNorm / upsampling on the channels-last view
            if not self._channels_last:
                for mod in (self.self_fusion_1x, self.self_fusion_2x, self.self_fusion_4x, self.fusion_down,
                            self.post_fusion_1, self.post_fusion_2, self.post_fusion_3, self.post_fusion_4):
                    mod.to(memory_format=torch.channels_last)
                self._channels_last = True
            feats_1x, feats_2x, feats_4x = (t.contiguous(memory_format=torch.channels_last)
                                            for t in (feats_1x, feats_2x, feats_4x))
        f1 = upsample2x_bilinear(self.self_fusion_1x(feats_1x))
        f2 = self.self_fusion_2x(feats_2x)
        f4 = self.pool4x(self.self_fusion_4x(feats_4x))
        x = self.fusion_down(torch.cat([f1, f2, f4], dim=1))
        for blk in (self.post_fusion_1, self.post_fusion_2, self.post_fusion_3, self.post_fusion_4):
            x = blk(x)
        return x

    def _feat_fusion_rows(self, feats_1x, feats_2x, feats_4x):
        """Inference on the GPU: the whole 2D stack on pixel-row matrices (channels-last), every
        convolution + BatchNorm on the HIP gather-GEMM path (dense2d.py); the 32-channel result is
        returned as a channels-last [V,32,H/8,W/8] view and feeds the back-projection in place."""
        return self._fusion_on_rows([D2.rows_of(t.float().contiguous(memory_format=torch.channels_last))
                                     for t in (feats_1x, feats_2x, feats_4x)], [tuple(t.shape) for t in (feats_1x, feats_2x, feats_4x)])

    def _fusion_on_rows(self, rows, shapes):
        """the stack on the pixel rows [V*h*w, C] of the three levels (shapes: their [V,C,h,w])"""
        dev = rows[0].device
        grids = [D2.PixelGrid.get(v, h, w, dev) for v, c, h, w in shapes]
        g1, g2, g4 = grids
        c1, c2, c4 = (r.shape[1] for r in rows)
        cat = torch.empty((g2.n, c1 + c2 + c4), dtype=torch.float32, device=dev)
        # The three per-level Fusion_Blocks are independent until the concat: the 1/16 level is only 85
        # row tiles (one wave per SIMD on a third of the CUs), so the levels run on three streams and
        # join at the concat (under HIP-graph capture this becomes three parallel branches).
        main = torch.cuda.current_stream()
        if self._side_streams is None:
            self._side_streams = (_lib.side_stream(dev, _lib.SIDE_BRANCH_A), _lib.side_stream(dev, _lib.SIDE_BRANCH_B))
        s1, s4 = self._side_streams
        # BatchNorm form (c) (round 6): the 36 layers of the pass leave their BatchNorm sums in accumulator blocks of ONE arena,
        # zeroed here by one fill in front of the fork; no finalize launch between two layers (EPRECON_BN_ACC=0: as before)
        if D2.BN_ACC and self._bn_arena is None:
            self._bn_arena = D2.BnArena(dev)
        with D2.bn_pass(self._bn_arena if D2.BN_ACC else None):
            s1.wait_stream(main)
            s4.wait_stream(main)
            with torch.cuda.stream(s4):
                f4 = self.self_fusion_4x.run_rows(rows[2], g4)
                cat[:, c1 + c2:] = D2.rows_of(self.pool4x(D2.maps_of(f4, g4.maps, g4.height, g4.width)))
            with torch.cuda.stream(s1):
                f1 = self.self_fusion_1x.run_rows(rows[0], g1)
                up = upsample2x_bilinear(D2.maps_of(f1, g1.maps, g1.height, g1.width))
                cat[:, 0:c1] = D2.rows_of(up)
            self.self_fusion_2x.run_rows(rows[1], g2, out=cat[:, c1:c1 + c2])
            main.wait_stream(s1)
            main.wait_stream(s4)
            a = self.fusion_down.run_act(D2.Act(cat), g2)
            for blk in (self.post_fusion_1, self.post_fusion_2, self.post_fusion_3, self.post_fusion_4):
                a = blk.run_act(a, g2)
            return D2.maps_of(D2.materialize(a), g2.maps, g2.height, g2.width)

    def _fusion_graphed(self, views):
        """feat_fusion_pre through a captured HIP graph (inference only; the result buffer is reused
        by the next call, the caller consumes it immediately).  `views`: three lists (1/16, 1/8, 1/4 level)
        of the per-view [C,h,w] maps; they are stacked straight into the graph's input buffers."""
        key = tuple(tuple(v[0].shape) + (len(v),) for v in views) + (views[0][0].device,)
        # The graph's inputs are the pixel ROWS of the three levels: one launch writes them from the 27 per-view maps
        # (eprecon_views_to_rows_async) where three torch.stack launches in front of the replay and three channels-last copies
        # inside it used to (HIP convolution path, float32 contiguous maps, <= 16 views; otherwise stacked NCHW inputs as before).
        direct = self.use_hip_conv and len(views[0]) <= 16 and all(
            m.dtype == torch.float32 and m.is_contiguous() for v in views for m in v)
        key = key + (direct,)
        entry = self._graphs.get(key)
        if entry is None:
            shapes = [(len(v),) + tuple(v[0].shape) for v in views]
            if direct:
                static_in = [torch.empty((n * h * w, c), dtype=torch.float32, device=views[0][0].device) for n, c, h, w in shapes]
                self._views_to_rows(views, static_in)
                run = lambda: self._fusion_on_rows(static_in, shapes)
            else:
                static_in = [torch.stack(v) for v in views]
                run = lambda: self.feat_fusion_pre(*static_in)
            side = _lib.side_stream(static_in[0].device, _lib.SIDE_SETUP)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):  # first-use work (kernel maps, packed weights) happens outside the capture
                    run()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of a multi-GPU run may touch the runtime during the capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                static_out = run()
            entry = (graph, static_in, static_out)
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        if direct:
            self._views_to_rows(views, static_in)
        else:
            for s_, v in zip(static_in, views):
                torch.stack(v, out=s_)
        graph.replay()
        return static_out

    @staticmethod
    def _views_to_rows(views, rows):
        """the per-view [C,h,w] maps of the three levels -> their pixel-row buffers, one launch"""
        import ctypes
        d = _lib.ViewsDesc()
        d.levels, d.n_views = len(views), len(views[0])
        for l, (v, r) in enumerate(zip(views, rows)):
            c, h, w = v[0].shape
            d.channels[l], d.hw[l], d.dst[l] = c, h * w, r.data_ptr()
            for i, m in enumerate(v):
                d.src[l][i] = m.data_ptr()
        _lib.check(_lib.load().eprecon_views_to_rows_async(ctypes.byref(d), _lib.current_stream()), "eprecon_views_to_rows_async")

    def sparse_stack(self, var, vset):
        """variance volume f32[N,32] on the voxel set -> occupancy logit f32[N,1]  (:131-174)"""
        x = self.norm0.run(var)
        x = self.similary_1.run(x, vset)
        for conv, norm in ((self.subm1, self.norm1), (self.subm2, self.norm2), (self.subm3, self.norm3)):
            x = conv.run_ln(x, vset, norm, relu=True, residual=x)  # LN(x + ReLU(conv(x))), one launch
        if torch.is_grad_enabled():
            return self.norm4.run(self.subm4.run(x, vset))
        y, partial = self.subm4.run_stats(x, vset)       # the logit layer writes norm4's summaries in its epilogue
        return self.norm4.run_partials(y, partial, out=y)

    def forward(self, coords, origin, voxel_size, features_all, KRcam, shape, stage, min_view_number, between=None):
        """`between` (optional, beyond the reference's signature): called once after the variance volume is queued and before
        its valid-voxel count is read on the host — work the caller queues there keeps the GPU busy while the host waits for
        the count and starts issuing the submanifold stack (inference on the GPU only; ignored otherwise)."""
        bs = features_all[0][0].shape[0]
        dev = features_all[0][0].device
        graphed = self.use_hip_graph and not torch.is_grad_enabled() and dev.type == "cuda"
        per_batch = []
        for b in range(bs):
            # per-view maps of batch element b: 1/16 [80,h,w], 1/8 [40,2h,2w], 1/4 [24,4h,4w]
            views = [[f[l][b] for f in features_all] for l in (2, 1, 0)]
            if graphed:
                fused_b = self._fusion_graphed(views)
                per_batch.append(fused_b if bs == 1 else fused_b.clone())
            else:
                per_batch.append(self.feat_fusion_pre(*[torch.stack(v) for v in views]))
        fused = per_batch[0].unsqueeze(1) if bs == 1 else torch.stack(per_batch, dim=1)  # [V,B,32,H,W]
        if between is not None and dev.type == "cuda" and not torch.is_grad_enabled():
            pend = BP.run_async(coords, origin, voxel_size, fused, KRcam, min_view_number, BP.MODE_VARIANCE,
                                min_valid_per_batch=INIT_MIN_VALID, want_mean=True)
            between()
            res = pend.result()
            if res is not None:
                res["var"] = res.pop("feats")
        else:
            res = BP.view_variance(coords, origin, voxel_size, fused, KRcam, min_view_number,
                                   min_valid=INIT_MIN_VALID)
        if res is None:
            return None
        interval = 2 ** (2 - stage)
        coord_valid = res["coords"]
        parts = []
        start = 0
        self.dense_checks = []   # (kept for callers that passed it on: a DenseMap defers its own off-grid check now, sparse.DenseMap)
        for b in range(bs):  # statistics of norm0 / norm4 are per batch element, as in the reference
            nb = res["n_valid_per_batch"][b]
            seg = slice(start, start + nb)
            # the valid voxels are a raster-ordered subset of the dense `shape` grid (generate_grid): well filled, so the
            # 3x3x3 layers of the stack take the dense-grid kernel (no hash grid, no kernel map)
            vset = SP.VoxelSet(coord_valid[seg], interval, dims=shape)
            parts.append(self.sparse_stack(res["var"][seg], vset))
            start += nb
        occ = parts[0] if bs == 1 else torch.cat(parts)
        out_coords = coord_valid if coords.dtype == torch.int32 else coord_valid.to(coords.dtype)
        return [occ, out_coords, res["count"]]
