"""numpy restatement of GRUFusion's map bookkeeping (models/gru_fusion.py:59-114,195-215,259-386 of
the reference) — test infrastructure.  Pinned against golden vectors captured from the
reference's own convert2dense / update_map (tests/golden/gru_fusion.npz).  The ConvGRU applied to
the gathered rows is passed in as `fuse` (oracle/spvcnn.convgru; identity in the pinned fixtures)."""
import numpy as np

F32 = np.float32


class ScaleState:
    """persistent sparse global map of one scale (+ its ground-truth twin)"""

    def __init__(self, channels, global_origin):
        self.C = np.zeros((0, 3), np.int64)
        self.F = np.zeros((0, channels), F32)
        self.tC = np.zeros((0, 3), np.int64)
        self.tF = np.zeros((0, 1), F32)
        self.origin = np.asarray(global_origin, F32)


def relative_origin(origin_partial, global_origin, voxel_size):
    """models/gru_fusion.py:292-293: fp32 divide, truncation toward zero (.long())"""
    q = (np.asarray(origin_partial, F32) - np.asarray(global_origin, F32)) / F32(voxel_size)
    return np.trunc(q).astype(np.int64)


def _inside(c, d):
    return ((c >= 0) & (c < d)).all(1)


def fuse_fragment(state, coords, values, origin_partial, tsdf_vol, occ_vol, interval, dim, base_voxel=0.04,
                  fuse=None):
    """one batch element of GRUFusion.forward in feature mode.  coords int[N,4] (b,x,y,z) finest
    units; tsdf_vol / occ_vol dense [dim]^3 ground truth of this scale (or None).
    Returns dict(updated int[N',3], values, global_values, tsdf_target, occ_target, valid)."""
    c = values.shape[1]
    rel = relative_origin(origin_partial, state.origin, base_voxel * interval)
    cur = np.floor_divide(np.asarray(coords)[:, 1:].astype(np.int64), interval)
    g_local = state.C - rel
    valid = _inside(g_local, dim)
    glob_vol = np.zeros((dim, dim, dim, c), F32)
    cur_vol = np.zeros((dim, dim, dim, c), F32)
    gl = g_local[valid]
    glob_vol[gl[:, 0], gl[:, 1], gl[:, 2]] = state.F[valid]
    cur_vol[cur[:, 0], cur[:, 1], cur[:, 2]] = values
    active = (glob_vol != 0).any(-1) | (cur_vol != 0).any(-1)
    updated = np.argwhere(active)
    vals = cur_vol[updated[:, 0], updated[:, 1], updated[:, 2]]
    gvals = glob_vol[updated[:, 0], updated[:, 1], updated[:, 2]]
    res = {"updated": updated, "values": vals, "global_values": gvals, "valid": valid, "rel": rel}
    tgt_vol = None
    if tsdf_vol is not None:
        t_local = state.tC - rel
        tvalid = _inside(t_local, dim)
        tgt_vol = np.ones((dim, dim, dim), F32)
        tl = t_local[tvalid]
        tgt_vol[tl[:, 0], tl[:, 1], tl[:, 2]] = state.tF[tvalid, 0]
        oc = np.argwhere(occ_vol)
        tgt_vol[oc[:, 0], oc[:, 1], oc[:, 2]] = tsdf_vol[occ_vol]      # current GT overwrites the map's
        res["tsdf_target"] = tgt_vol[updated[:, 0], updated[:, 1], updated[:, 2]][:, None]
        res["occ_target"] = np.abs(res["tsdf_target"]) < 1
    new_vals = fuse(gvals, vals, updated, rel) if fuse is not None else vals
    res["fused"] = new_vals
    # update_map (:195-215)
    state.F = np.concatenate([state.F[~valid], new_vals])
    state.C = np.concatenate([state.C[~valid], updated + rel])
    if tgt_vol is not None:
        keep = np.abs(tgt_vol) < 1
        state.tF = np.concatenate([state.tF[~tvalid], tgt_vol[keep][:, None]])
        state.tC = np.concatenate([state.tC[~tvalid], np.argwhere(keep) + rel])
    return res
