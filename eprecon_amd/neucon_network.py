"""Coarse-to-fine 3D path — mirror of NeuConNet (models/neucon_network.py:25-624 of the reference).
Under torch.no_grad() this is the fused inference path; with autograd enabled the same forward runs through the
recording operators (eprecon_amd/autograd.py) and fills loss_dict like the reference: the initial-occupancy loss
(only_train_init), the per-scale TSDF / occupancy losses against the fused ground truth and the panoptic set
criterion (eprecon_amd/criterion.py).  Target visualisation is out of scope.

forward(features, features_backbone2d_occ_pano, inputs, outputs, ...) keeps the reference's
signature and early-return conventions:
  A  occupancy initialisation on the dense 48^3 grid -> stage-0 voxels           (:239-342)
  B  for i in 0..2: [upsample] -> Back_Project -> SPVCNN -> GRU fusion -> TSDF / occupancy heads
     -> sparsify                                                                  (:348-511)
  C  panoptic: ancestor pruning of the coarser levels, 48-channel projections, mask features
     (three submanifold residual blocks), mask-transformer decoder (PyTorch-ROCm) + panoptic
     post-processing                                                              (:516-587)
Every sparse / gather / scatter step runs in libeprecon_hip.so; the dense heads are PyTorch-ROCm.
"""
import sys

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import grid_ops as GO
from . import sparse as SP
from .back_project import Back_Project
from .backbone import stack_views
from .criterion import HungarianMatcher, SetCriterion, compute_loss, compute_loss_init
from .config import (CH_IMG, CH_INIT_DOWN, CH_VOXEL, EXCEED_NUM, INIT_MIN_VIEW, INIT_OCC_THRESHOLD, INIT_STAGE,
                     N_VIEWS, NUM_CLASSES, NUM_QUERIES, PANOPTIC_CH, PANOPTIC_SHAPE, STAGE_MIN_OCC)
from .generate_grids import dense_coords
from .gru_fusion import GRUFusion
from .mask3dformer import MultiScaleMaskedTransformerDecoder, panoptic_post
from .modules import Linear4xTrans, Panoptic_Feat_Fusion, SPVCNN, linear4x_pair
from .occupancy_initialization import Occupancy_Initialization
from .tensor import PointTensor
from .torchsparse_utils import SpvcnnPrefetch, aligned_camera_coords

# the reference's sequence of torch calls (threshold, index_add counts, nonzero, index_select x 4, cat: ~15 launches and two
# host reads per level) stays for training and the seeded random drop; inference takes eprecon_sparsify_async (3 launches, one
# host read).  Tests flip this module attribute to compare the two.
_FUSED_SPARSIFY = True
# EPRECON_PREFETCH=0: every SPVCNN pass reads the sizes of its voxel sets itself and the panoptic pruning reads its own counts
# (round 4: 14 blocking reads per fragment) instead of queueing that work on the DEVICE count of the rows a compaction has just
# written and letting its counts ride on the compaction's read (torchsparse_utils.SpvcnnPrefetch: 10 reads).  Same results.
_PREFETCH = __import__("os").environ.get("EPRECON_PREFETCH", "1") == "1"


WARN_TAG = ""    # prefix of the guard warnings below; fragment_step.calibrate_occupancy_heads sets "[calibration] " around its forwards


def _warn(msg):
    print(f"[eprecon_amd] {WARN_TAG}warning: {msg}", file=sys.stderr)


class NeuConNet(nn.Module):
    def __init__(self, cfg, panoptic_decoder="default"):
        super().__init__()
        self.cfg = cfg
        self.n_scales = len(cfg.THRESHOLDS) - 1
        alpha = int(cfg.ALPHA)
        # channels entering each SPVCNN: image features (+ previous stage's features, tsdf, occ)
        ch_in = [80 * alpha, 96 + 40 * alpha + 2, 48 + 24 * alpha + 2, 24 + 24 + 2]
        channels = list(CH_VOXEL)
        gru_channels = [a + b for a, b in zip(channels, CH_IMG)]
        self.channels = channels
        self.back_projection = nn.ModuleList()
        if cfg.FUSION.FUSION_ON:
            self.gru_fusion = GRUFusion(cfg, ch_in=gru_channels, ch_voxel=channels)
        self.sp_convs = nn.ModuleList()
        self.tsdf_preds = nn.ModuleList()
        self.occ_preds = nn.ModuleList()
        self.panoptic_preds = nn.ModuleList()
        self.initialization = Occupancy_Initialization(CH_IMG, CH_INIT_DOWN, N_VIEWS)
        self.panoptic_feat_fusion = Panoptic_Feat_Fusion(channels[2], PANOPTIC_CH, CH_IMG)
        if panoptic_decoder == "default":  # models/neucon_network.py:59-71
            panoptic_decoder = MultiScaleMaskedTransformerDecoder(
                mask_classification=True, num_classes=NUM_CLASSES, hidden_dim=PANOPTIC_CH, num_queries=NUM_QUERIES,
                nheads=8, dim_feedforward=4 * PANOPTIC_CH, dec_layers=6, pre_norm=False, mask_dim=PANOPTIC_CH)
        self.panoptic = panoptic_decoder  # None skips the decoder (outputs['panoptic_levels'] only)
        for i in range(len(cfg.THRESHOLDS)):
            self.back_projection.append(Back_Project(CH_IMG[i]))
            self.sp_convs.append(SPVCNN(num_classes=1, in_channels=ch_in[i], pres=1, cr=1 / 2 ** i,
                                        vres=cfg.VOXEL_SIZE * 2 ** (self.n_scales - i),
                                        dropout=cfg.SPARSEREG_DROPOUT))
            self.tsdf_preds.append(Linear4xTrans(channels[i], 1))
            self.occ_preds.append(Linear4xTrans(channels[i], 1))
            self.panoptic_preds.append(Linear4xTrans(gru_channels[i], PANOPTIC_CH))
        # criterion (models/neucon_network.py:73-99): class / mask / dice weights 0.2 / 0.8 / 0.8, one copy per decoder layer
        class_w, mask_w, dice_w = 0.2, 0.8, 0.8
        weight_dict = {"loss_ce": class_w, "loss_mask": mask_w, "loss_dice": dice_w}
        for j in range(6):
            weight_dict.update({f"{k}_{j}": v for k, v in (("loss_ce", class_w), ("loss_mask", mask_w), ("loss_dice", dice_w))})
        self.criterion = SetCriterion(NUM_CLASSES, HungarianMatcher(class_w, mask_w, dice_w), weight_dict, eos_coef=0.1,
                                      losses=["labels", "masks"])
        self.trace = None  # set to a list to record per-stage intermediates (parity tests)
        self.panoptic_stream = None        # a torch.cuda.Stream: pipelined serving (forward, section C)
        self.panoptic_worker = None        # a concurrent.futures.ThreadPoolExecutor(1): issues that branch's launches
        self.distributed_exchange = False  # multi-GPU: boundary-voxel all-gather before each fragment

    @staticmethod
    def _lookup(volume, coords, scale):
        c = coords.long()
        return volume[c[:, 0], c[:, 1] // 2 ** scale, c[:, 2] // 2 ** scale, c[:, 3] // 2 ** scale]

    @torch.no_grad()
    def get_target(self, coords, inputs, scale):
        """models/neucon_network.py:117-126 (fusion off): ground truth at the voxels"""
        return self._lookup(inputs["tsdf_list"][scale], coords, scale), self._lookup(inputs["occ_list"][scale], coords, scale)

    @torch.no_grad()
    def get_target_init(self, coords, inputs, scale):
        """models/neucon_network.py:128-143: (clamp(1 - |tsdf|, 0, 1), occ) at the voxels"""
        tsdf, occ = self.get_target(coords, inputs, scale)
        return torch.clamp(1 - tsdf.abs(), min=0, max=1), occ

    @torch.no_grad()
    def get_panoptic_targets(self, coords, inputs, scale, bs):
        """models/neucon_network.py:157-191: per batch element the instance masks over its voxels and, per instance, the
        most frequent semantic label"""
        semantic = self._lookup(inputs["semantic_list"][scale], coords, scale)
        instance = self._lookup(inputs["instance_list"][scale], coords, scale)
        targets = []
        for b in range(bs):
            rows = coords[:, 0] == b
            sem, ins = semantic[rows].long(), instance[rows]
            ids, inv = torch.unique(ins, return_inverse=True)
            masks = inv.unsqueeze(0) == torch.arange(ids.shape[0], device=ins.device).unsqueeze(1)
            votes = torch.zeros((ids.shape[0], int(sem.max()) + 1 if sem.numel() else 1), dtype=torch.int64, device=ins.device)
            votes.index_put_((inv, sem), torch.ones_like(sem), accumulate=True)
            targets.append({"labels": votes.argmax(1), "masks": masks})
        return targets

    # models/neucon_network.py:193-214
    def upsample(self, pre_feat, pre_coords, interval, num=8):
        assert num == 8
        return GO.upsample(pre_feat, pre_coords, interval)

    def _record(self, **kw):
        if self.trace is not None:
            self.trace.append(kw)

    def _init_stage(self, features, inputs, bs, dev, select=True):
        """stage A of forward (:239-318): occupancy initialisation on the dense 48^3 grid and the stage-0 voxel selection ->
        (init_output | None, coord_init_selected | None, shape_init)"""
        cfg = self.cfg
        interval = 2 ** (self.n_scales - INIT_STAGE)
        scale = self.n_scales - INIT_STAGE
        up_coords, shape_init = dense_coords(cfg.N_VOX, interval, bs, device=dev)
        KRcam = inputs["proj_matrices"][:, :, scale].permute(1, 0, 2, 3).contiguous()
        init_output = self.initialization(up_coords, inputs["vol_origin_partial"], cfg.VOXEL_SIZE, features,
                                          KRcam, shape_init, INIT_STAGE, INIT_MIN_VIEW)
        if init_output is None or not select:
            return init_output, None, shape_init
        occ_init, coord_init, _ = init_output
        # (the same host read checks that every voxel of the initialisation set was on the dense grid its convolutions ran on)
        selected, _ = GO.init_select(occ_init, coord_init, bs, dim=shape_init[0] // 2 ** INIT_STAGE,
                                     cell=2 ** self.n_scales, threshold=INIT_OCC_THRESHOLD,
                                     must_be_zero=getattr(self.initialization, "dense_checks", ()))
        return init_output, selected, shape_init

    def _spvcnn_behind(self, level, inputs, children):
        """the hook (grid_ops.sparsify / back_project.forward_behind) that queues the coordinate side of SPVCNN pass `level`
        on the device count of the rows just compacted; finish -> (up_coords | None, r_coords)"""
        cfg, net = self.cfg, self.sp_convs[level]
        interval = 2 ** (self.n_scales - level)
        res = float(net.vres) / float(net.pres) if net.pres != 1 else float(net.vres)     # initial_voxelize's resolution

        def behind(coords, n_dev, extra_out):
            pf = SpvcnnPrefetch(coords, n_dev, children, interval, inputs["vol_origin_partial"], cfg.VOXEL_SIZE,
                                inputs["world_to_aligned_camera"], res, summary=extra_out)
            return lambda m, host: pf.finish(8 * m if children else m, host)
        behind.n_extra = 6       # (status, unique count) of the three strided voxel sets
        return behind

    @staticmethod
    def _prune_behind(coords1, coords0):
        """the ancestor pruning of levels 1 and 0 (prune_to_ancestors) queued on the device count of the finest level's kept
        rows; finish -> (keep1, keep0, n1, n0)"""
        def behind(fine, n_dev, extra_out):
            dev = fine.device
            keep1 = SP.HashGrid(fine.shape[0], dev).build(fine, quantum=2, n_dev=n_dev).query(coords1.contiguous()) >= 0
            keep0 = SP.HashGrid(fine.shape[0], dev).build(fine, quantum=4, n_dev=n_dev).query(coords0.contiguous()) >= 0
            extra_out.copy_(torch.stack([keep1.sum(), keep0.sum()]))
            return lambda m, host: (keep1, keep0, int(host[0]), int(host[1]))
        behind.n_extra = 2       # the two kept-row counts
        return behind

    def forward(self, features, features_backbone2d_occ_pano, inputs, outputs, only_train_init=False,
                only_train_occ=False, init_overlap_count=0):
        cfg = self.cfg
        bs = features[0][0].shape[0]
        dev = features[0][0].device
        loss_dict = {}
        recording = torch.is_grad_enabled()
        # the reference's placeholder losses stay attached to the graph (0 * features.sum(), :255,377)
        zero = features[0][0].sum() * 0.0 if recording and features[0][0].requires_grad else torch.zeros((), device=dev)
        if self.distributed_exchange and cfg.FUSION.FUSION_ON:
            # the one collective of the path (RCCL all-gather of boundary voxels); placed
            # before every data-dependent early return so that all ranks issue it once per fragment
            for b in range(bs):
                self.gru_fusion.exchange_boundaries(inputs, b)

        # ---- A. occupancy initialisation ("depth prior") -------------------------------------
        scale = self.n_scales - INIT_STAGE
        init_output, coord_init_selected, shape_init = self._init_stage(features, inputs, bs, dev, select=not only_train_init)
        if init_output is None:
            loss_dict["occupancy_initialization_loss"] = zero
            _warn("no valid points in initialization")
            outputs["init_overlap_count"] = init_overlap_count
            return outputs, loss_dict
        occ_init, coord_init, count_init = init_output
        if only_train_init:    # :267-292
            tsdf_init_target, occ_init_target = self.get_target_init(coord_init, inputs, scale)
            hit = occ_init.detach().sigmoid().reshape(-1) > INIT_OCC_THRESHOLD
            losses = []
            for b in range(bs):
                rows = coord_init[:, 0] == b
                pred, tgt = hit[rows], tsdf_init_target[rows] > 0
                init_overlap_count = init_overlap_count + (pred & tgt).sum() / (pred | tgt).sum()
                losses.append(compute_loss_init(occ_init[rows], tsdf_init_target[rows], occ_init_target[rows]))
            outputs["init_overlap_count"] = init_overlap_count
            loss_dict["occupancy_initialization_loss"] = sum(losses) / len(losses)
            return outputs, loss_dict
        self._record(stage="init", occ_init=occ_init, coord_init=coord_init, selected=coord_init_selected)

        # ---- B. coarse-to-fine surface reconstruction ----------------------------------------
        pre_feat = pre_coords = None
        panoptic_voxel_feats, panoptic_coords = [], []
        occ_target = occupancy = None
        # inference on the GPU: the coordinate side of every SPVCNN pass (and the panoptic pruning) is queued on the device
        # count of the rows the previous compaction wrote, and its sizes ride on that compaction's host read
        prefetch = _PREFETCH and dev.type == "cuda" and not recording and _FUSED_SPARSIFY
        ahead = pruned = None        # (up_coords, r_coords) of the level about to run; (keep1, keep0, n1, n0)
        for i in range(cfg.N_LAYER):
            interval = 2 ** (self.n_scales - i)
            scale = self.n_scales - i
            ready, ahead = ahead, None
            if i == 0:
                up_coords = coord_init_selected.contiguous()
                min_view_number = 2
                up_feat = None
            else:
                if ready is not None and ready[0] is not None and ready[0].shape[0] == 8 * pre_coords.shape[0]:
                    up_feat, up_coords = GO.upsample(pre_feat, pre_coords, interval, up_coords=ready[0])
                else:
                    ready = None
                    up_feat, up_coords = self.upsample(pre_feat, pre_coords, interval)
                min_view_number = 0
            feats = stack_views([f[scale] for f in features_backbone2d_occ_pano])   # no copy for batched backbones
            KRcam = inputs["proj_matrices"][:, :, scale].permute(1, 0, 2, 3).contiguous()
            if i == 0 and prefetch and up_coords.dtype == torch.int32:
                from . import back_project as BP
                project_output, ready = BP.forward_behind(self.back_projection[0], up_coords, inputs["vol_origin_partial"],
                                                          cfg.VOXEL_SIZE, feats, KRcam, min_view_number,
                                                          self._spvcnn_behind(0, inputs, children=False))
            else:
                project_output = self.back_projection[i](up_coords, inputs["vol_origin_partial"], cfg.VOXEL_SIZE,
                                                         feats, KRcam, min_view_number)
            if project_output is None:
                loss_dict[f"tsdf_occ_loss_{i}"] = zero
                _warn(f"no valid points in back_projection: scale {i}")
                return outputs, loss_dict
            volume, up_coords, _, _, count = project_output
            if i != 0:
                if min_view_number <= 0:       # (the view count is never negative: every voxel is kept, no host read)
                    feat = torch.cat([volume, up_feat], dim=1)
                else:
                    keep = count >= min_view_number
                    feat = torch.cat([volume, up_feat if bool(keep.all()) else up_feat[keep]], dim=1)
            else:
                feat = volume

            if ready is not None and ready[1].shape[0] == up_coords.shape[0]:
                r_coords = ready[1]        # (queued ahead with the voxelisation the SPVCNN pass below finds in the cache)
            else:
                r_coords = aligned_camera_coords(up_coords, inputs["vol_origin_partial"], cfg.VOXEL_SIZE,
                                                 inputs["world_to_aligned_camera"])
            sp_in = feat
            feat = self.sp_convs[i](PointTensor(feat.contiguous(), r_coords))
            feat_all = torch.cat([feat, volume], dim=-1)
            self._record(stage=f"spvcnn{i}", coords=up_coords, feat_in=sp_in, r_coords=r_coords, feat_out=feat,
                         volume=volume)

            tsdf_target = None
            if not cfg.FUSION.FUSION_ON and "occ_list" in inputs:
                tsdf_target, occ_target = self.get_target(up_coords, inputs, scale)
            if cfg.FUSION.FUSION_ON:
                voxel_dim = feat.shape[-1]
                fuse_in = (up_coords, feat_all)
                up_coords, feat_all, tsdf_target, occ_target = self.gru_fusion(up_coords, feat_all, inputs, i)
                feat = feat_all[:, :voxel_dim]
                self._record(stage=f"gru{i}", coords_in=fuse_in[0], feat_in=fuse_in[1], coords=up_coords,
                             feat_all=feat_all, tsdf_target=tsdf_target)
            tsdf, occ = linear4x_pair(self.tsdf_preds[i], self.occ_preds[i], feat)
            if recording and tsdf_target is not None:   # :441-449 (grid_mask is all ones with FUSION.FULL)
                loss_dict[f"tsdf_occ_loss_{i}"] = compute_loss(tsdf, occ, tsdf_target, occ_target, pos_weight=cfg.POS_WEIGHT)
            else:
                loss_dict[f"tsdf_occ_loss_{i}"] = zero

            # ---- sparsify for the next stage (:454-507) ----
            # (the reference's grid_mask is all ones on this path, `occupancy[grid_mask == False] = False` is a no-op)
            # Inference on the GPU: threshold, the guards' counts and the compaction of the kept rows in ONE call and one host
            # read (csrc/grid_ops.hip, eprecon_sparsify_async); the random sub-sampling branch and training keep the
            # reference's sequence of torch calls below.
            fused = None
            if _FUSED_SPARSIFY and dev.type == "cuda" and not recording and feat_all.stride(1) == 1:
                behind = None
                if prefetch and i + 1 < cfg.N_LAYER:
                    behind = self._spvcnn_behind(i + 1, inputs, children=True)
                elif prefetch and i + 1 == cfg.N_LAYER and len(panoptic_coords) == 2 and bs == 1 \
                        and all(c.dtype == torch.int32 for c in panoptic_coords):
                    behind = self._prune_behind(panoptic_coords[1], panoptic_coords[0])
                fused = GO.sparsify(occ, cfg.THRESHOLDS[i], occ_target,
                                    up_coords if up_coords.dtype == torch.int32 else up_coords.to(torch.int32), tsdf, feat_all,
                                    feat.shape[1], bs, behind=behind)
                if self.training and any(cfg.TRAIN_NUM_SAMPLE[i] < nb <= cfg.TRAIN_NUM_SAMPLE[i] * EXCEED_NUM
                                         for nb in fused[0][1:1 + bs]):
                    fused = None          # a batch element is over its cap: np.random.choice drops rows first (below)
                elif behind is not None:
                    if i + 1 < cfg.N_LAYER:
                        ahead = fused[6]
                    else:
                        pruned = fused[6]
            if fused is not None:
                stats = [fused[0][1:1 + bs], fused[0][1 + bs:1 + 2 * bs]]
                if self.trace is not None:
                    self._record(stage=f"heads{i}", feat=feat, tsdf=tsdf, occ=occ, occupancy=occ.squeeze(1) > cfg.THRESHOLDS[i])
                for b in range(bs):
                    if stats[0][b] < STAGE_MIN_OCC:
                        _warn(f"no valid points: scale {i}")
                        return outputs, loss_dict
                    if self.training and stats[0][b] > cfg.TRAIN_NUM_SAMPLE[i] * EXCEED_NUM:
                        _warn(f"exceed too many points: scale {i} num_batch {stats[0][b]}")
                        return outputs, loss_dict
                if occ_target is not None:
                    for b in range(bs):
                        if stats[1][b] == 0:
                            _warn(f"occ_target is 0: scale {i}")
                            return outputs, loss_dict
                _, pre_coords, pre_tsdf, pre_occ, kept_all, pre_feat = fused[:6]
                if pre_coords.dtype != up_coords.dtype:
                    pre_coords = pre_coords.to(up_coords.dtype)
                panoptic_voxel_feats.append(kept_all)
                panoptic_coords.append(pre_coords)
                if i == cfg.N_LAYER - 1:
                    outputs["coords"] = pre_coords
                    outputs["tsdf"] = pre_tsdf
                continue
            occupancy = occ.squeeze(1) > cfg.THRESHOLDS[i]
            # recorded before the guards so that an early return still leaves the logits visible;
            # `occupancy` is the same tensor object the sub-sampling below edits in place
            self._record(stage=f"heads{i}", feat=feat, tsdf=tsdf, occ=occ, occupancy=occupancy)
            # per batch element: occupied voxels, and occupied voxels whose target is occupied — ONE host read
            # for all the guards below (the reference synchronises once per guard and batch element)
            bcol = up_coords[:, 0].long()
            tgt = occ_target.reshape(-1) if occ_target is not None else torch.ones_like(occupancy)   # [N] (get_target) or [N,1] (fusion)
            count_by_batch = lambda: torch.zeros((2, bs), dtype=torch.int64, device=dev).index_add_(
                1, bcol, torch.stack([occupancy, occupancy & tgt]).long()).tolist()
            stats = count_by_batch()
            dropped = False
            for b in range(bs):
                num_batch = stats[0][b]
                if num_batch < STAGE_MIN_OCC:
                    _warn(f"no valid points: scale {i}")
                    return outputs, loss_dict
                cap = cfg.TRAIN_NUM_SAMPLE[i]
                # self.training is True on the reference's test path (main.py:357): the caps are live
                if self.training and num_batch > cap * EXCEED_NUM:
                    _warn(f"exceed too many points: scale {i} num_batch {num_batch}")
                    return outputs, loss_dict
                elif self.training and num_batch > cap:
                    _warn(f"choice too many points: scale {i} num_batch {num_batch}")
                    batch_ind = torch.nonzero(up_coords[:, 0] == b).squeeze(1)
                    choice = np.random.choice(num_batch, num_batch - cap, replace=False)
                    ind = torch.nonzero(occupancy[batch_ind]).squeeze(1)
                    drop = batch_ind[ind[torch.from_numpy(choice).to(ind.device)]]
                    occupancy[drop] = False
                    dropped = True
            if occ_target is not None:
                if dropped:
                    stats = count_by_batch()
                for b in range(bs):
                    if stats[1][b] == 0:
                        _warn(f"occ_target is 0: scale {i}")
                        return outputs, loss_dict
            keep_rows = torch.nonzero(occupancy).squeeze(1)
            pre_coords = up_coords.index_select(0, keep_rows)
            pre_tsdf, pre_occ = tsdf.index_select(0, keep_rows), occ.index_select(0, keep_rows)
            kept_all = feat_all.index_select(0, keep_rows)
            panoptic_voxel_feats.append(kept_all)
            panoptic_coords.append(pre_coords)
            pre_feat = torch.cat([kept_all[:, :feat.shape[1]], pre_tsdf, pre_occ], dim=1)
            if i == cfg.N_LAYER - 1:
                outputs["coords"] = pre_coords
                outputs["tsdf"] = pre_tsdf

        # ---- C. panoptic segmentation (:516-587) ---------------------------------------------------
        side = self.panoptic_stream if (not recording and dev.type == "cuda" and self.panoptic is not None) else None
        if side is None:
            self._panoptic_branch(panoptic_coords, panoptic_voxel_feats, bs, outputs, pruned)
            if self.panoptic is not None:
                outputs["panoptic_info"] = [panoptic_post(o) for o in outputs["panoptic_out"]]  # :583-587
                if recording and "rgb_list" in inputs and occ_target is not None:
                    loss_dict["panoptic_loss"] = self._panoptic_loss(outputs["panoptic_out"], panoptic_coords[2], occ_target,
                                                                     occupancy, inputs, bs)
        else:
            # Pipelined serving (opt-in, NeuConNet.panoptic_stream): the panoptic branch of THIS fragment is queued on its
            # own stream and the call returns; its ~400 small launches then overlap with the front of the NEXT fragment
            # (it reads only this fragment's tensors, never the scene map).  The host-synchronising post-processing is
            # deferred: outputs["panoptic_finish"]() waits for the branch, fills outputs["panoptic_info"] and returns outputs.
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            for t in list(panoptic_voxel_feats) + list(panoptic_coords) + (list(pruned[:2]) if pruned is not None else []):
                t.record_stream(side)       # allocated on the main stream, read on the side stream after this call returns
            def branch(outputs=outputs, coords=panoptic_coords, feats=panoptic_voxel_feats, side=side, pruned=pruned):
                # (grad mode and the current stream are thread-local: both are set here)
                with torch.no_grad(), torch.cuda.stream(side):
                    self._panoptic_branch(coords, feats, bs, outputs, pruned)
                    if self.panoptic_worker is not None:
                        # the host-synchronising post-processing too: it is the worker that waits for the decoder
                        outputs["panoptic_info"] = [panoptic_post(o) for o in outputs["panoptic_out"]]
                    return side.record_event()

            # With a worker thread the ~400 launches of the branch are issued while the main thread sits in the blocking
            # count reads of the NEXT fragment (the GIL is released there); without one they are issued inline.
            job = self.panoptic_worker.submit(branch) if self.panoptic_worker is not None else None
            done = None if job is not None else branch()

            def finish(outputs=outputs, job=job, done=done, side=side):
                (job.result() if job is not None else done).synchronize()
                if "panoptic_info" not in outputs:      # inline mode: deferred to here so that forward() did not block
                    with torch.cuda.stream(side):
                        outputs["panoptic_info"] = [panoptic_post(o) for o in outputs["panoptic_out"]]
                    side.synchronize()
                outputs.pop("panoptic_finish", None)
                return outputs
            outputs["panoptic_finish"] = finish
        self._record(stage="panoptic", coords=panoptic_coords, feats=panoptic_voxel_feats)
        if recording:
            _lib.drain_deferred()   # training has no fused sparsify read: the back-projections' deferred row-count checks are verified here
        return outputs, loss_dict

    def _panoptic_branch(self, panoptic_coords, panoptic_voxel_feats, bs, outputs, pruned=None):
        """ancestor pruning of the coarser levels, 48-channel projections, mask features, decoder (:516-581); fills
        outputs['panoptic_levels'] and outputs['panoptic_out'] on the current stream.  pruned = (keep1, keep0, n1, n0): the
        pruning was queued ahead and its counts came with the finest level's sparsify read (_prune_behind)"""
        if pruned is not None:
            keep1, keep0, n1, n0 = pruned
        else:
            keep1, keep0 = self.prune_to_ancestors(panoptic_coords)
            # (one nonzero per level, shared by the coordinates and the features: boolean indexing runs it once per tensor)
            # both counts in ONE read, then the row lists with their sizes given (torch.nonzero sizes its result on the host:
            # two reads)
            _lib.count_host_read()
            n1, n0 = torch.stack([keep1.sum(), keep0.sum()]).tolist()
        i1 = torch.nonzero_static(keep1, size=n1).squeeze(1)
        i0 = torch.nonzero_static(keep0, size=n0).squeeze(1)
        panoptic_coords[1], panoptic_voxel_feats[1] = panoptic_coords[1].index_select(0, i1), panoptic_voxel_feats[1].index_select(0, i1)
        panoptic_coords[0], panoptic_voxel_feats[0] = panoptic_coords[0].index_select(0, i0), panoptic_voxel_feats[0].index_select(0, i0)
        for p in range(3):
            panoptic_voxel_feats[p] = self.panoptic_preds[p](panoptic_voxel_feats[p])
        outputs["panoptic_levels"] = []
        panoptic_predictions = []
        for b in range(bs):
            if bs == 1:     # every row belongs to batch element 0: no row selection (3 x nonzero + 8 gathers + 3 host reads)
                sel = lambda t, p: t
            else:
                rows = [torch.nonzero(panoptic_coords[p][:, 0] == b).squeeze(1) for p in range(3)]
                sel = lambda t, p, rows=rows: t[rows[p]]
            c2 = sel(panoptic_coords[2], 2)
            mask_features = self.panoptic_feat_fusion.generate_mask_features(
                panoptic_feats=sel(panoptic_voxel_feats[2], 2), coords_b=torch.zeros_like(c2[:, 0]),
                coords_xyz=c2[:, 1:], batch_size=1, spitial_shape=PANOPTIC_SHAPE)
            feats_b = [sel(panoptic_voxel_feats[p], p).unsqueeze(0).permute(0, 2, 1) for p in range(3)]
            coords_b = [sel(panoptic_coords[p], p)[..., 1:].unsqueeze(0) for p in range(3)]
            outputs["panoptic_levels"].append({"features": feats_b, "coords": coords_b,
                                               "mask_features": mask_features})
            if self.panoptic is not None:
                out_b = self.panoptic(panoptic_features=feats_b, panoptic_coords=coords_b,
                                      mask_features=mask_features.unsqueeze(0).permute(0, 2, 1),
                                      spitial_shape=PANOPTIC_SHAPE)
                panoptic_predictions.append(out_b)
        if self.panoptic is not None:
            outputs["panoptic_out"] = panoptic_predictions

    def _panoptic_loss(self, panoptic_outs, coords_fine, occ_target, occupancy, inputs, bs):
        """models/neucon_network.py:589-622: the set criterion on the voxels whose ground truth is observed; the
        weighted terms are summed and divided by 3, then averaged over the batch"""
        # (the supervised voxels as index lists found once: boolean-mask indexing finds them again — a blocking read — for each
        # of the seven prediction heads, and once more per head in the backward; criterion.take)
        from .criterion import mask_rows, take
        supervised = take(occ_target.view(-1), mask_rows(occupancy)).view(-1)
        sup_rows = mask_rows(supervised)
        for b in range(bs):
            keep = sup_rows if bs == 1 else mask_rows(supervised[coords_fine[:, 0] == b])
            panoptic_outs[b]["pred_masks"] = take(panoptic_outs[b]["pred_masks"], keep, -1)
            for aux in panoptic_outs[b]["aux_outputs"]:
                aux["pred_masks"] = take(aux["pred_masks"], keep, -1)
        targets = self.get_panoptic_targets(coords_fine[sup_rows], inputs, 0, bs)
        total = []
        for b in range(bs):
            losses = self.criterion(panoptic_outs[b], [targets[b]])
            total.append(sum(v * self.criterion.weight_dict[k] for k, v in losses.items() if k in self.criterion.weight_dict) / 3)
        return sum(total) / len(total)

    @staticmethod
    def prune_to_ancestors(panoptic_coords):
        """models/neucon_network.py:516-542 (K17): keep the level-1 voxels that are floor(c/2)*2 of some
        level-2 voxel, and the level-0 voxels that are floor(c/4)*4 of one — a hash-grid membership
        query instead of the reference's [N1, M, 4] broadcast compare."""
        fine = panoptic_coords[2].contiguous()
        dev = fine.device
        g2 = SP.HashGrid(fine.shape[0], dev).build(fine, quantum=2)
        keep1 = g2.query(panoptic_coords[1].contiguous()) >= 0
        g4 = SP.HashGrid(fine.shape[0], dev).build(fine, quantum=4)
        keep0 = g4.query(panoptic_coords[0].contiguous()) >= 0
        return keep1, keep0
