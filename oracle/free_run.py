"""FREE-RUNNING CPU restatement of NeuConNet.forward for a batch of B >= 1 fragment windows of ONE scene with an empty scene map
(models/neucon_network.py:230-511 of the reference; B > 1: its `for b in range(bs)` loops, BatchNorm statistics over the voxels of
all windows, GRU fusion window by window on the shared map, models/gru_fusion.py:275) — test infrastructure, never imported by
the product.

oracle/neucon.py checks the HIP path stage by stage on the inputs the HIP path fed each stage (teacher forcing), because
a near-zero occupancy logit may legitimately flip between fp32 implementations.  This module chains the same oracle
pieces end to end instead: every stage consumes what the previous ORACLE stage produced.  It also records, per stage, the
margin |occupancy logit| of every voxel, so that a test can tell a legitimate flip (margin below the fp32 noise) from a
real difference.  Parity unpinned like the pieces it is built from (torchsparse / spconv semantics, oracle/sparse.py).

  A  occupancy initialisation on the dense interval-2 grid -> stage-0 voxels      (:239-318, occupancy_initialization.py:61-182)
  B  for i in 0..2: [upsample] -> Back_Project -> SPVCNN -> GRU fusion (empty map: h = 0) -> heads -> sparsify   (:348-511)
"""
import numpy as np

from . import back_project as OB
from . import grid_ops as OG
from . import gru_fusion as OGF
from . import neucon as ONC
from . import occupancy_init as OI
from . import pointvoxel as PV
from . import spvcnn as ON

F32 = np.float32
CH_VOXEL = (96, 48, 24)
CH_ALL = (176, 88, 48)


def calibrate(occ, keep_fraction):
    """the rule of eprecon_amd.fragment_step.calibrate_occupancy_heads: logit' = (logit - q) / sigma"""
    q = np.quantile(occ.astype(np.float64), 1.0 - keep_fraction)
    sigma = max(float(occ.astype(np.float64).std(ddof=1)), 1e-12)
    return F32(q), F32(sigma)


def forward(sd, fused_init, feats2, inputs, n_vox=(96, 96, 96), voxel_size=0.04, keep_fraction=None, caps=(15000, 60000, 120000)):
    """sd: NeuConNet.state_dict() as numpy (occupancy heads are rescaled IN PLACE when keep_fraction is given);
    fused_init f32[V,B,32,h,w]: the output of Occupancy_Initialization.feat_fusion_pre per window (dense 2D convolutions, run by
    the caller with the PyTorch modules on the CPU); feats2: backbone #2 pyramid, list over views of [f4, f8, f16] (each [B,C,H,W]);
    inputs: the numpy dict of eprecon_amd.synthetic.make_model_inputs.  Returns a dict of per-stage records."""
    origin = inputs["vol_origin_partial"]
    w2ac = inputs["world_to_aligned_camera"]
    bs = origin.shape[0]
    rec = {"stages": []}
    # ---- A ----
    coords48 = _dense(n_vox, 2, bs)
    kr1 = np.ascontiguousarray(inputs["proj_matrices"][:, :, 1].transpose(1, 0, 2, 3))
    r = OB.back_project(coords48, origin, voxel_size, fused_init, kr1, 2, OB.MODE_VARIANCE)
    init_sd = ONC.sub_dict(sd, "initialization")
    # the submanifold stack runs per window (batch_size 1, its BatchNorms see one window's voxels: the stack sits INSIDE the
    # reference's `for batch in range(bs)` loop, models/occupancy_initialization.py:79-176)
    logit = np.concatenate([OI.sparse_stack(init_sd, r["feats"][r["coords"][:, 0] == b], r["coords"][r["coords"][:, 0] == b], 2)
                            for b in range(bs)])
    selected = OG.init_select(logit, r["coords"], bs)
    rec["init"] = {"n_valid": len(r["coords"]), "n_selected": len(selected),
                   "sigmoid_margin": np.abs(1.0 / (1.0 + np.exp(-logit[:, 0].astype(np.float64))) - 0.3)}
    # ---- B ----
    pre_feat = pre_coords = None
    for i in range(3):
        interval, scale = 2 ** (2 - i), 2 - i
        if i == 0:
            up_coords, up_feat, min_view = selected, None, 2
        else:
            up_feat, up_coords = OG.upsample(pre_feat, pre_coords, interval)
            min_view = 0
        f = np.stack([v[scale] for v in feats2])                                   # [V,B,C,H,W]
        kr = np.ascontiguousarray(inputs["proj_matrices"][:, :, scale].transpose(1, 0, 2, 3))
        bp = OB.back_project(up_coords, origin, voxel_size, f, kr, min_view)
        volume, coords = bp["feats"], bp["coords"]
        if i != 0:
            keep = bp["count"] >= min_view
            feat = np.concatenate([volume, up_feat[keep]], 1)
        else:
            feat = volume
        _, feat = ONC.spvcnn_stage(sd, i, coords, feat, origin, w2ac, voxel_size)
        feat_all = np.concatenate([feat, volume], 1)
        # GRU fusion, window by window on ONE map per scale (empty before the first window): the union is the window's voxels
        # with a non-zero row and the map's voxels inside its volume, raster order; the second window of a batch sees what the
        # first one fused (models/gru_fusion.py:275-389)
        state = OGF.ScaleState(CH_ALL[i], inputs["vol_origin"][0])
        cv, vres = CH_VOXEL[i], voxel_size * interval
        fused_c, fused_f = [], []
        for b in range(bs):
            def fuse(gvals, vals, updated, rel, b=b, i=i, interval=interval, cv=cv, vres=vres):
                # (the reference's aligned-camera points of the fusion carry batch column 0, :332-337)
                c4 = np.concatenate([np.zeros((len(updated), 1), np.int32), (updated * interval).astype(np.int32)], 1)
                pts = PV.aligned_coords(c4, origin[b:b + 1], voxel_size, w2ac[b:b + 1])
                fv = ON.convgru(sd, f"gru_fusion.fusion_nets_voxel.{i}", gvals[:, :cv], vals[:, :cv], pts, 1, vres)
                fi = ON.convgru(sd, f"gru_fusion.fusion_nets_img.{i}", gvals[:, cv:], vals[:, cv:], pts, 1, vres)
                return np.concatenate([fv, fi], 1)

            rows = coords[:, 0] == b
            g = OGF.fuse_fragment(state, coords[rows], feat_all[rows], origin[b], None, None, interval, n_vox[0] // interval,
                                  base_voxel=voxel_size, fuse=fuse)
            fused_c.append(np.concatenate([np.full((len(g["updated"]), 1), b, np.int32), (g["updated"] * interval).astype(np.int32)], 1))
            fused_f.append(g["fused"])
        coords, feat_all = np.concatenate(fused_c), np.concatenate(fused_f)
        feat = feat_all[:, :cv]
        if keep_fraction is not None:   # rescale this stage's occupancy head on the oracle's own logits
            _, occ_raw, _ = ONC.heads_stage(sd, i, feat)
            q, sigma = calibrate(occ_raw[:, 0], keep_fraction[i])
            sd[f"occ_preds.{i}.linear3.bias"] = ((sd[f"occ_preds.{i}.linear3.bias"] - q) / sigma).astype(F32)
            sd[f"occ_preds.{i}.linear3.weight"] = (sd[f"occ_preds.{i}.linear3.weight"] / sigma).astype(F32)
        tsdf, occ, occupancy = ONC.heads_stage(sd, i, feat)
        n_occ = int(occupancy.sum())
        per_batch = [int(occupancy[coords[:, 0] == b].sum()) for b in range(bs)]
        rec["stages"].append({"coords": coords, "occ": occ[:, 0].copy(), "tsdf": tsdf[:, 0].copy(), "occupancy": occupancy.copy(),
                              "n_in": len(up_coords), "n_fused": len(coords), "n_occ": n_occ, "n_occ_per_batch": per_batch})
        if min(per_batch) < 500 or max(per_batch) > caps[i]:     # the reference's guards per batch element (:469-484): the synthetic windows must stay inside them
            rec["early"] = i
            return rec
        pre_coords = coords[occupancy]
        pre_feat = np.concatenate([feat[occupancy], tsdf[occupancy], occ[occupancy]], 1)
    rec["coords"], rec["tsdf"] = pre_coords, pre_feat[:, -2:-1].copy()
    return rec


def _dense(n_vox, interval, batch=1):
    g, _ = OG.generate_grid(n_vox, interval)       # f32[3, n] x-major
    rows = [np.concatenate([np.full((1, g.shape[1]), b, F32), g]).T for b in range(batch)]
    return np.ascontiguousarray(np.concatenate(rows).astype(np.int32))
