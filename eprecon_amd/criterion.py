"""Training losses of the panoptic head and of the TSDF / occupancy heads — mirror of models/criterion.py
(SetCriterion :85-296, dice_loss :20-39, sigmoid_ce_loss :41-65), models/matcher.py (HungarianMatcher :51-147) and the
static loss helpers of NeuConNet (models/neucon_network.py:627-700, utils.py apply_log_transform).  SURVEY.md 8f row 4.

Dense PyTorch (autograd does the backward); the Hungarian assignment runs on the host with scipy like the reference's.
Pinned against the reference's own modules on seeded inputs (tests/golden/criterion.npz).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

# ScanNet ids of the 20 evaluated classes; position + 1 is the class index the network predicts (0 = no object)
VALID_CLASS_IDS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39)
MIN_MASK_VOXELS = 100          # masks with <= 100 voxels are dropped as noise (models/criterion.py:239-246)
MAX_MASK_POS_WEIGHT = 30.0


class _Take(torch.autograd.Function):
    """x.index_select(dim, idx) for a list of DISTINCT indices (the rows a mask keeps: idx = mask.nonzero()).  What boolean-mask
    indexing computes — but the list is found once and shared by every tensor the mask is applied to (x[mask] runs nonzero,
    a blocking device -> host read of the row count, per tensor), and the backward writes the gradient rows into zeros with
    index_copy_ (IndexBackward0 re-runs nonzero on the mask: a second blocking read per tensor).  Same values, bit for bit:
    every destination row receives exactly one source row.  One optimisation step had 172 blocking reads, 38 of them inside
    the backward (tools/profile_train_syncs.py)."""

    @staticmethod
    def forward(ctx, x, idx, dim):
        ctx.save_for_backward(idx)
        ctx.shape, ctx.dim = x.shape, dim
        return x.index_select(dim, idx)

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        out = grad.new_zeros(ctx.shape)
        out.index_copy_(ctx.dim, idx, grad.contiguous())
        return out, None, None


def take(x, idx, dim=0):
    """rows / columns `idx` (distinct, int64, on x's device) of x along dim"""
    if dim < 0:
        dim += x.dim()
    return _Take.apply(x, idx, dim) if x.requires_grad else x.index_select(dim, idx)


def mask_rows(mask):
    """the row list of a boolean mask: ONE blocking read (its length)"""
    return torch.nonzero(mask.reshape(-1)).squeeze(1)


def apply_log_transform(tsdf):
    """sign(t) * log(|t| + 1)  (utils.py apply_log_transform)"""
    return torch.sign(tsdf) * torch.log(torch.abs(tsdf) + 1)


def compute_pos_weight(targets):
    """#negatives / #positives of a {0,1} target"""
    flat = targets.reshape(-1)
    n_pos = flat.sum()
    return (flat.shape[0] - n_pos).float() / n_pos


def compute_loss(tsdf, occ, tsdf_target, occ_target, loss_weight=(1, 1), mask=None, pos_weight=1.0):
    """NeuConNet.compute_loss (models/neucon_network.py:667-700): occupancy BCE with the class-balance weight
    (#neg / #pos * pos_weight) + L1 of the log-transformed TSDF on the occupied targets"""
    tsdf, occ, tsdf_target, occ_target = tsdf.reshape(-1), occ.reshape(-1), tsdf_target.reshape(-1), occ_target.reshape(-1)
    if mask is not None:
        rows = mask_rows(mask)
        tsdf, occ, tsdf_target, occ_target = take(tsdf, rows), take(occ, rows), tsdf_target[rows], occ_target[rows]
    pos = mask_rows(occ_target)      # (its length is n_pos: the guard below costs no second read)
    if pos.numel() == 0:
        return tsdf.sum() * 0.0
    n_pos = occ_target.sum()
    w = (occ_target.shape[0] - n_pos).float() / n_pos * pos_weight
    occ_loss = F.binary_cross_entropy_with_logits(occ, occ_target.float(), pos_weight=w)
    tsdf_loss = (apply_log_transform(take(tsdf, pos)) - apply_log_transform(tsdf_target[pos])).abs().mean()
    return loss_weight[0] * occ_loss + loss_weight[1] * tsdf_loss


def compute_loss_init(occ_init, tsdf_init_target, occ_init_target):
    """NeuConNet.compute_loss_init (models/neucon_network.py:627-664): BCE of the initial occupancy logits against
    (tsdf_target > 0) on the voxels whose target is observed"""
    occ_init, t, o = occ_init.reshape(-1), tsdf_init_target.reshape(-1), occ_init_target.reshape(-1)
    rows = mask_rows((t == 0) | (o == 1))
    occ_init, t = take(occ_init, rows), t[rows]
    if t.sum() == 0:
        return occ_init.sum() * 0.0
    target = (t > 0).float()
    return F.binary_cross_entropy_with_logits(occ_init, target, pos_weight=compute_pos_weight(target))


def dice_loss(inputs, targets, num_masks):
    p = inputs.sigmoid().flatten(1)
    num = 2 * (p * targets).sum(-1)
    den = p.sum(-1) + targets.sum(-1)
    return (1 - (num + 1) / (den + 1)).sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks):
    """per matched mask: class-balanced BCE (pos_weight = #neg / #pos clamped to 30) averaged over its voxels; mean over masks"""
    # (one pass over the [T, N] block with a per-row weight instead of a Python loop of T reductions: the same means)
    flat_t = targets.flatten(1)
    n_pos = flat_t.sum(1)
    w = torch.clamp((flat_t.shape[1] - n_pos).float() / n_pos, max=MAX_MASK_POS_WEIGHT)
    per_mask = F.binary_cross_entropy_with_logits(inputs.flatten(1), flat_t, pos_weight=w[:, None], reduction="none").mean(1)
    return per_mask.sum() / per_mask.shape[0]


class HungarianMatcher(nn.Module):
    """1-to-1 assignment of predicted queries to ground-truth masks minimising
    cost_mask * BCE + cost_class * (-p[class]) + cost_dice * dice   (models/matcher.py:75-119)"""

    def __init__(self, cost_class=1.0, cost_mask=1.0, cost_dice=1.0):
        super().__init__()
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0
        self.cost_class, self.cost_mask, self.cost_dice = cost_class, cost_mask, cost_dice

    @torch.no_grad()
    def forward_many(self, heads, targets):
        """forward() for several prediction heads over the SAME targets (the decoder's final output and its auxiliary
        layers, models/criterion.py:269-283): the cost matrices of all heads from batched products and ONE device -> host
        transfer instead of one per head; the assignments themselves stay scipy's, one per head -> [forward(h, targets) ...]"""
        n_heads = len(heads)
        costs = []
        for b in range(heads[0]["pred_logits"].shape[0]):
            prob = torch.stack([h["pred_logits"][b] for h in heads]).softmax(-1)            # [H, Q, K + 1]
            c_class = -prob[:, :, targets[b]["labels"]]
            logits = torch.stack([h["pred_masks"][b] for h in heads]).float()               # [H, Q, N]
            gt = targets[b]["masks"].to(logits)
            n_vox = logits.shape[2]
            pos = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits), reduction="none")
            neg = F.binary_cross_entropy_with_logits(logits, torch.zeros_like(logits), reduction="none")
            c_mask = (pos @ gt.t() + neg @ (1 - gt).t()) / n_vox
            p = logits.sigmoid()
            c_dice = 1 - (2 * p @ gt.t() + 1) / (p.sum(-1)[..., None] + gt.sum(-1)[None, None, :] + 1)
            costs.append(self.cost_mask * c_mask + self.cost_class * c_class + self.cost_dice * c_dice)
        host = [c.cpu() for c in costs] if len(costs) > 1 else [costs[0].cpu()]
        pairs = []
        for cost in host:
            for h in range(n_heads):
                pairs.append(linear_sum_assignment(cost[h].reshape(cost.shape[1], -1)))
        # the assignments go back to the device in ONE transfer (the losses index device tensors with them: as host tensors
        # every such indexing — three per head and loss — was a blocking copy of its own)
        import numpy as np
        flat = torch.as_tensor(np.concatenate([np.asarray(a, dtype=np.int64) for ij in pairs for a in ij]) if pairs else
                               np.zeros(0, np.int64)).to(costs[0].device)
        out = [[] for _ in range(n_heads)]
        at = 0
        for n, (i, j) in enumerate(pairs):
            out[n % n_heads].append((flat[at:at + len(i)], flat[at + len(i):at + len(i) + len(j)]))
            at += len(i) + len(j)
        return out

    @torch.no_grad()
    def forward(self, outputs, targets):
        out = []
        for b in range(outputs["pred_logits"].shape[0]):
            prob = outputs["pred_logits"][b].softmax(-1)
            c_class = -prob[:, targets[b]["labels"]]
            logits = outputs["pred_masks"][b].float()
            gt = targets[b]["masks"].to(logits)
            n_vox = logits.shape[1]
            pos = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits), reduction="none")
            neg = F.binary_cross_entropy_with_logits(logits, torch.zeros_like(logits), reduction="none")
            c_mask = (pos @ gt.t() + neg @ (1 - gt).t()) / n_vox
            p = logits.sigmoid()
            c_dice = 1 - (2 * p @ gt.t() + 1) / (p.sum(-1)[:, None] + gt.sum(-1)[None, :] + 1)
            cost = self.cost_mask * c_mask + self.cost_class * c_class + self.cost_dice * c_dice
            i, j = linear_sum_assignment(cost.reshape(prob.shape[0], -1).cpu())
            out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
        return out


class SetCriterion(nn.Module):
    """models/criterion.py:85-296.  forward(outputs, targets) with outputs {'pred_logits' [B,Q,K+1], 'pred_masks' [B,Q,N],
    'aux_outputs': [...]} and targets [{'labels' int64[T] (ScanNet ids), 'masks' bool[T,N]}] (batch 1, like the reference)
    -> dict of losses ('loss_ce', 'loss_mask', 'loss_dice' and '<name>_<i>' for the auxiliary heads)."""

    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict, self.eos_coef, self.losses = num_classes, matcher, weight_dict, eos_coef, losses
        w = torch.ones(num_classes + 1)
        w[0] = eos_coef
        self.register_buffer("empty_weight", w)
        self.register_buffer("valid_classes", torch.tensor(VALID_CLASS_IDS, dtype=torch.int64), persistent=False)

    @staticmethod
    def _perm(indices, which):
        batch = torch.cat([torch.full_like(pair[which], i) for i, pair in enumerate(indices)])
        return batch, torch.cat([pair[which] for pair in indices])

    def loss_labels(self, outputs, targets, indices, num_masks):
        logits = outputs["pred_logits"].float()
        matched = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
        if len(matched) == 0:
            return {"loss_ce": logits.sum() * 0.0}
        target = torch.zeros(logits.shape[:2], dtype=torch.int64, device=logits.device)
        target[self._perm(indices, 0)] = matched
        return {"loss_ce": F.cross_entropy(logits.transpose(1, 2), target, self.empty_weight)}

    def loss_masks(self, outputs, targets, indices, num_masks):
        src = outputs["pred_masks"][self._perm(indices, 0)]
        tgt = targets[0]["masks"].unsqueeze(0).to(src)[self._perm(indices, 1)]
        if len(src) == 0 or len(tgt) == 0:   # ("loss_masks" is the reference's key on this branch, models/criterion.py:160-164)
            return {"loss_masks": src.sum() * 0.0, "loss_dice": src.sum() * 0.0}
        return {"loss_mask": sigmoid_ce_loss(src, tgt, num_masks), "loss_dice": dice_loss(src, tgt, num_masks)}

    def _filter_targets(self, outputs, targets):
        """keep the 20 evaluated classes (re-indexed 1..20) and the voxels they cover, then drop masks of <= 100 voxels
        and the voxels only those covered; the predictions are restricted to the surviving voxels (:206-252)"""
        t = targets[0]
        for min_voxels in (None, MIN_MASK_VOXELS):
            if min_voxels is None:
                keep = torch.isin(t["labels"], self.valid_classes.to(t["labels"].device))
            else:
                keep = t["masks"].sum(1) > min_voxels
            # (the kept masks and the voxels they cover as index lists, found once each: boolean indexing finds them again —
            # a blocking read — for every tensor it is applied to, and once more per tensor in the backward)
            kept = mask_rows(keep)
            masks = t["masks"][kept]
            cols = mask_rows(masks.any(0)) if kept.numel() else kept
            labels = t["labels"][kept]
            if min_voxels is None:   # ScanNet id -> 1 + position in the list of evaluated classes
                labels = torch.searchsorted(self.valid_classes.to(labels.device), labels) + 1
            t["labels"], t["masks"] = labels, masks[:, cols]
            if cols.numel() == 0:    # (no kept mask covers a voxel: the selected block sums to zero)
                return False
            outputs["pred_masks"] = take(outputs["pred_masks"], cols, -1)
            for aux in outputs.get("aux_outputs", []):
                aux["pred_masks"] = take(aux["pred_masks"], cols, -1)
        return True

    def forward(self, outputs, targets):
        if not self._filter_targets(outputs, targets):
            return {"loss_dice": outputs["pred_logits"].sum() * 0.0}
        main = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        aux_list = list(outputs.get("aux_outputs", []))
        if hasattr(self.matcher, "forward_many"):     # all heads matched behind one device -> host transfer
            matched = self.matcher.forward_many([main] + aux_list, targets)
        else:
            matched = [self.matcher(h, targets) for h in [main] + aux_list]
        indices = matched[0]
        num_masks = float(max(sum(len(t["labels"]) for t in targets), 1))
        fns = {"labels": self.loss_labels, "masks": self.loss_masks}
        losses = {}
        for name in self.losses:
            losses.update(fns[name](outputs, targets, indices, num_masks))
        for i, aux in enumerate(aux_list):
            idx = matched[1 + i]
            for name in self.losses:
                losses.update({f"{k}_{i}": v for k, v in fns[name](aux, targets, idx, num_masks).items()})
        return losses
