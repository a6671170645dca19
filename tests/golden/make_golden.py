"""Regenerates tests/golden/*.npz by running the REFERENCE's own Python functions
(/root/reference, imported through ref_shim) on seeded synthetic inputs.

Run in the build container only:   python tests/golden/make_golden.py [name ...]

Inputs are never stored: tests regenerate them from the recorded seeds with
eprecon_amd.synthetic (numpy default_rng, identical on every machine).  Stored outputs are
reduced to keep each fixture small: full integer results (visible-view counts, which determine
the valid set and the output order), plus sampled feature rows and per-row checksums.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shim  # noqa: E402

torch = ref_shim.install()

from eprecon_amd import synthetic as S  # noqa: E402


def _save(name, **arrays):
    arrays["torch_version"] = np.array(torch.__version__)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def _sample_rows(n, k, seed):
    rng = np.random.default_rng(seed)
    if n <= k:
        return np.arange(n, dtype=np.int64)
    return np.sort(rng.choice(n, size=k, replace=False)).astype(np.int64)


def bp_case(window, lvl, interval, n_vox, feat_seed, min_view, batch=1):
    c, h, w = S.pyramid_shapes(*window["image_hw"])[lvl]
    feats = S.make_features(feat_seed, 9, (c, h, w), batch=batch)
    coords = S.dense_coords(n_vox, interval, batch=batch)
    kr = np.repeat(window["proj_matrices"][:, lvl][:, None], batch, axis=1).copy()
    origin = np.repeat(window["vol_origin_partial"][None], batch, axis=0).copy()
    if batch > 1:  # make the second batch element see a shifted volume
        origin[1:, 0] += 0.36
    return coords, origin, feats, np.ascontiguousarray(kr)


def gen_back_project():
    from models.occupancy_initialization import Back_Project
    from ops.back_project import back_project as ops_back_project

    out = {}
    # cfg1: 320x240 window, dense 32^3 single-scale volume on the 1/4-res map [24,60,80]
    # cfg2: 640x480 window, dense 24^3 / 48^3 / 96^3 on the three pyramid levels
    cases = [("cfg1_l0", dict(seed=1, width=320, height=240, n_vox=(32, 32, 32)), 0, 1, 1),
             ("cfg1b2_l1", dict(seed=2, width=320, height=240, n_vox=(32, 32, 32)), 1, 2, 2),
             ("cfg2_l2", dict(seed=0), 2, 4, 1),
             ("cfg2_l1", dict(seed=0), 1, 2, 1),
             ("cfg2_l0", dict(seed=0), 0, 1, 1)]
    for name, wargs, lvl, interval, batch in cases:
        window = S.make_window(**wargs)
        for mv in (0, 2):
            coords, origin, feats, kr = bp_case(window, lvl, interval, window["n_vox"],
                                                 1000 + lvl, mv, batch)
            tc, to, tf, tk = (torch.from_numpy(x) for x in (coords, origin, feats, kr))
            res = Back_Project(feats.shape[2])(tc, to, window["voxel_size"], tf, tk, mv)
            vol, vcoords, grid, mask, count = [r.numpy() for r in res]
            n_valid = vol.shape[0]
            rows = _sample_rows(n_valid, 384, 7)
            key = f"{name}_mv{mv}"
            out[key + "_count"] = count.astype(np.uint8)
            out[key + "_nvalid"] = np.int64(n_valid)
            out[key + "_rows"] = rows
            out[key + "_feat_rows"] = vol[rows]
            out[key + "_grid_rows"] = grid[:, rows]
            out[key + "_mask_rows"] = mask[:, rows]
            out[key + "_coord_rows"] = vcoords[rows].astype(np.int32)
            out[key + "_rowsum"] = vol.sum(axis=1, dtype=np.float64).astype(np.float32)[::32]
            if mv == 2 and name.startswith("cfg1"):
                # canonical op with the normalised-depth channel (ops/back_project.py:69-75)
                r2 = ops_back_project(tc, to, window["voxel_size"], tf, tk, mv)
                out[key + "_depth_rows"] = r2[0].numpy()[rows, -1]
                assert np.array_equal(r2[0].numpy()[:, :-1], vol)
        out[name + "_meta"] = np.array([lvl, interval, batch, wargs.get("seed", 0),
                                        wargs.get("width", 640), wargs.get("height", 480),
                                        window["n_vox"][0], 1000 + lvl], dtype=np.int64)
    _save("back_project", **out)


def gen_grid_ops():
    """generate_grid, NeuConNet.upsample and the init -> coarse selection
    (models/neucon_network.py:193-228,298-318) on seeded inputs"""
    import torch.nn.functional as F
    from ops.generate_grids import generate_grid
    from models.neucon_network import NeuConNet

    out = {}
    for interval in (1, 2, 4):
        g, dims = generate_grid([96, 96, 96], interval)
        out[f"grid_i{interval}_checksum"] = np.array([float(g.numpy().astype(np.float64).sum()),
                                                      float((g.numpy().astype(np.float64) *
                                                             np.arange(1, g.shape[1] + 1)).sum())])
        out[f"grid_i{interval}_head"] = g.numpy()[:, :200]
        out[f"grid_i{interval}_dims"] = np.array(dims)
    rng = np.random.default_rng(31)
    coords = rng.integers(0, 24, size=(500, 3)) * 4
    coords = np.concatenate([rng.integers(0, 2, size=(500, 1)), coords], 1).astype(np.int32)
    feat = rng.standard_normal((500, 7)).astype(np.float32)
    uf, uc = NeuConNet.upsample(None, torch.from_numpy(feat), torch.from_numpy(coords), 2)
    out["up_feat"], out["up_coords"] = uf.numpy(), uc.numpy()
    # selection: seeded logits on the voxels of a 48^3 grid seen by >= 2 views (here: a random
    # 85 % subset), smooth blob structure so that erosion leaves something
    gx, gy, gz = np.meshgrid(*[np.arange(48)] * 3, indexing="ij")
    blob = (np.sin(gx / 5.0) + np.cos(gy / 7.0) + np.sin(gz / 6.0 + 1.0)).astype(np.float32)
    noise = rng.standard_normal((48, 48, 48)).astype(np.float32) * 0.6
    logit_vol = blob + noise - 0.4
    valid = rng.random((48, 48, 48)) < 0.85
    sel_out = []
    shape_init = (48, 48, 48)
    logit = torch.from_numpy(logit_vol[valid])          # raster order of the valid voxels
    occ_sel = logit.sigmoid() > 0.3
    vol = torch.zeros(shape_init, dtype=torch.bool)
    vol[torch.from_numpy(valid)] = occ_sel
    vol = F.max_pool3d(vol.unsqueeze(0).float(), 2).squeeze(0)
    vol = NeuConNet.erode(None, vol, kernel_size=3)
    vol = NeuConNet.dilate(None, vol, kernel_size=3)
    vol = NeuConNet.dilate(None, vol, kernel_size=3)
    nz = torch.nonzero(vol).squeeze(1) * 4
    out["sel_logit_vol"] = logit_vol
    out["sel_valid"] = valid
    out["sel_coords"] = nz.numpy().astype(np.int32)
    _save("grid_ops", **out)


def gen_dense_blocks():
    """the reference's dense PyTorch blocks (models/modules.py:273-399) with seeded weights:
    state_dict + input seed + output, so that eprecon_amd.modules can be checked key-for-key"""
    from models.modules import Conv2d_Residual_Block, Fusion_Block, Linear4xTrans, Linear_Residual

    out = {}
    torch.manual_seed(123)
    cases = {"fusion8": (Fusion_Block(8), (9, 8, 12, 16)),
             "res6": (Conv2d_Residual_Block(6, 3), (9, 6, 10, 10)),
             "l4x_12_1": (Linear4xTrans(12, 1), (50, 12)),
             "l4x_12_12": (Linear4xTrans(12, 12), (50, 12)),
             "linres10": (Linear_Residual(10), (40, 10))}
    for name, (mod, shape) in cases.items():
        mod.train()
        with torch.no_grad():
            for p in mod.parameters():
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.2)
        x = torch.from_numpy(np.random.default_rng(5).standard_normal(shape).astype(np.float32))
        sd = {k: v.clone() for k, v in mod.state_dict().items()}  # before BN running-stat updates
        with torch.no_grad():
            y = mod(x)
        for k, v in sd.items():
            out[f"{name}__sd__{k}"] = v.numpy()
        out[f"{name}__shape"] = np.array(shape)
        out[f"{name}__out"] = y.numpy()
    _save("dense_blocks", **out)


def gru_sequence_inputs(scale, n_frag=3, seed=77, n_vox=24, ch=(6, 4, 3)):
    """seeded fragment sequence for the GRU-fusion bookkeeping fixtures (shared with the tests)"""
    rng = np.random.default_rng(seed + scale)
    interval = 2 ** (2 - scale)
    d = n_vox // interval
    c = ch[scale]
    shifts = [(0, 0, 0), (8, 0, 0), (8, 8, -8), (0, 16, 0)][:n_frag]
    frags = []
    for k, sh in enumerate(shifts):
        occ = rng.random((d, d, d)) < 0.25
        xyz = np.argwhere(occ)
        rng.shuffle(xyz)
        xyz = xyz[: max(8, len(xyz) // 2)]
        vals = rng.standard_normal((len(xyz), c)).astype(np.float32)
        vals[rng.random(len(xyz)) < 0.15] = 0.0          # all-zero rows: the "inactive voxel" quirk
        tsdf = np.clip(rng.standard_normal((d, d, d)) * 0.8, -1, 1).astype(np.float32)
        occ_gt = (np.abs(tsdf) < 0.999) & (rng.random((d, d, d)) < 0.5)
        frags.append({"coords": np.concatenate([np.zeros((len(xyz), 1), np.int64), xyz * interval], 1).astype(np.int32),
                      "values": vals, "tsdf": tsdf, "occ": occ_gt,
                      "origin_partial": (np.array([-0.96, 0.2, -0.4]) + np.array(sh) * 0.04).astype(np.float32)})
    return frags, np.array([-0.96, 0.2, -0.4], np.float32), interval, d, c


def gen_gru_fusion():
    """GRUFusion.convert2dense / update_map (models/gru_fusion.py:67-114,195-215) driven through the
    glue of GRUFusion.forward (:275-331,366-386) with the ConvGRU replaced by identity (the ConvGRU
    needs torchsparse).  Three overlapping fragments per scale."""
    from types import SimpleNamespace
    import models.gru_fusion as G

    class PT:
        def __init__(self, F, C):
            self.F, self.C = F, C

        def cuda(self):
            return self

        def detach(self):
            return self

    G.PointTensor = PT
    cfg = SimpleNamespace(THRESHOLDS=[0, 0, 0], VOXEL_SIZE=0.04, N_VOX=[24, 24, 24], N_LAYER=3,
                          FUSION=SimpleNamespace(FULL=True))
    out = {}
    for scale in (1, 2):
        fus = G.GRUFusion(cfg, ch_in=[6, 4, 3], ch_voxel=[4, 3, 2])
        frags, global_origin, interval, d, c = gru_sequence_inputs(scale)
        fus.reset(scale)
        fus.global_origin[scale] = torch.from_numpy(global_origin)
        for k, fr in enumerate(frags):
            voxel_size = cfg.VOXEL_SIZE * interval
            rel = ((torch.from_numpy(fr["origin_partial"]) - fus.global_origin[scale]) / voxel_size).long()
            coords = torch.from_numpy(fr["coords"])
            coords_b = torch.div(coords[:, 1:].long(), interval, rounding_mode="floor")
            values = torch.from_numpy(fr["values"])
            occ_t = torch.from_numpy(fr["occ"])
            tsdf_t = torch.from_numpy(fr["tsdf"])[occ_t]
            coords_t = torch.nonzero(occ_t)
            upd, cur_vol, glob_vol, tgt_vol, valid, valid_t = fus.convert2dense(coords_b, values, coords_t, tsdf_t, rel, scale)
            vals = cur_vol[upd[:, 0], upd[:, 1], upd[:, 2]]
            gvals = glob_vol[upd[:, 0], upd[:, 1], upd[:, 2]]
            tsdf_u = tgt_vol[upd[:, 0], upd[:, 1], upd[:, 2]]
            fus.update_map(vals, upd, tgt_vol, valid, valid_t, rel, scale)   # identity fusion
            key = f"s{scale}_f{k}_"
            out[key + "updated"] = upd.numpy().astype(np.int32)
            out[key + "values"] = vals.numpy()
            out[key + "global_values"] = gvals.numpy()
            out[key + "valid"] = valid.numpy()
            out[key + "tsdf_target"] = tsdf_u.numpy()
            out[key + "rel"] = rel.numpy()
            out[key + "map_C"] = fus.global_volume[scale].C.numpy().astype(np.int32)
            out[key + "map_F"] = fus.global_volume[scale].F.numpy()
            out[key + "tgt_C"] = fus.target_tsdf_volume[scale].C.numpy().astype(np.int32)
            out[key + "tgt_F"] = fus.target_tsdf_volume[scale].F.numpy()
    _save("gru_fusion", **out)


def mask3d_inputs(seed=11, c=16):
    """three voxel levels on the 4-grid (every finest voxel coincides with a coarser voxel, so the
    reference's float cdist/argmin has no ties to break arbitrarily)"""
    rng = np.random.default_rng(seed)
    cells = rng.permutation(np.argwhere(np.ones((10, 10, 10), bool)))[:320] * 4
    c2 = cells[:300]
    c1 = rng.permutation(cells)[:220]
    c1 = np.unique(np.concatenate([c1, c2[:150]]), axis=0)
    c0 = np.unique(np.concatenate([rng.permutation(cells)[:90], c2[100:200]]), axis=0)
    feats = [rng.standard_normal((1, c, len(x))).astype(np.float32) for x in (c0, c1, c2)]
    mask_feat = rng.standard_normal((1, c, len(c2))).astype(np.float32)
    return [c0, c1, c2], feats, mask_feat


def gen_mask3dformer():
    """MultiScaleMaskedTransformerDecoder.forward + panoptic_post (models/mask3dformer.py:198-581)"""
    from models.mask3dformer import MultiScaleMaskedTransformerDecoder, panoptic_post

    torch.manual_seed(5)
    dec = MultiScaleMaskedTransformerDecoder(mask_classification=True, num_classes=20, hidden_dim=16, num_queries=12,
                                             nheads=4, dim_feedforward=64, dec_layers=4, pre_norm=False, mask_dim=16)
    with torch.no_grad():
        dec.class_embed.bias[1:8] += 1.5       # make some queries confident about real classes
        dec.class_embed.weight.mul_(3.0)
    coords, feats, mask_feat = mask3d_inputs()
    out = {}
    for k, v in dec.state_dict().items():
        out["sd__" + k] = v.numpy()
    with torch.no_grad():
        res = dec([torch.from_numpy(f) for f in feats], [torch.from_numpy(c)[None] for c in coords],
                  torch.from_numpy(mask_feat), (40, 40, 40))
        post = panoptic_post({"pred_logits": res["pred_logits"], "pred_masks": res["pred_masks"]})
    out["pred_logits"] = res["pred_logits"].numpy()
    out["pred_masks"] = res["pred_masks"].numpy()
    out["aux_last_masks"] = res["aux_outputs"][-1]["pred_masks"].numpy()
    out["panoptic_seg"] = post["panoptic_seg"][0].numpy()
    info = post["panoptic_seg"][1]
    out["segments"] = np.array([[d["id"], int(d["isthing"]), d["category_id"]] for d in info], np.int64).reshape(-1, 3)
    _save("mask3dformer", **out)



def gen_mask3dformer_at_size():
    """the decoder in its production configuration (models/neucon_network.py:59-71: 80 queries, 48 channels, 8 heads, 6 layers)
    on >= 20k voxels per level: what the graph-replayed / SDPA path of eprecon_amd/mask3dformer.py is tuned for.  Stored:
    the state_dict, logits in full, the mask logits at 2,048 sampled voxels + per-query float64 sums, the panoptic labels."""
    from models.mask3dformer import MultiScaleMaskedTransformerDecoder, panoptic_post

    torch.manual_seed(9)
    dec = MultiScaleMaskedTransformerDecoder(mask_classification=True, num_classes=20, hidden_dim=48, num_queries=80,
                                             nheads=8, dim_feedforward=192, dec_layers=6, pre_norm=False, mask_dim=48)
    with torch.no_grad():
        dec.class_embed.bias[1:8] += 1.5
        dec.class_embed.weight.mul_(3.0)
    from cases import mask3d_inputs_at_size
    coords, feats, mask_feat = mask3d_inputs_at_size()
    out = {"n_per_level": np.array([len(c) for c in coords])}
    for k, v in dec.state_dict().items():
        out["sd__" + k] = v.numpy()
    with torch.no_grad():
        res = dec([torch.from_numpy(f) for f in feats], [torch.from_numpy(c)[None] for c in coords],
                  torch.from_numpy(mask_feat), (96, 96, 96))
        post = panoptic_post({"pred_logits": res["pred_logits"], "pred_masks": res["pred_masks"]})
    masks = res["pred_masks"][0].numpy()
    cols = _sample_rows(masks.shape[1], 2048, 99)
    out["pred_logits"] = res["pred_logits"].numpy()
    out["mask_cols"] = cols
    out["pred_masks_sampled"] = masks[:, cols]
    out["pred_masks_rowsum"] = masks.astype(np.float64).sum(1)
    out["aux_last_masks_sampled"] = res["aux_outputs"][-1]["pred_masks"][0].numpy()[:, cols]
    out["panoptic_seg"] = post["panoptic_seg"][0].numpy()
    # margin of the per-voxel decision, so that the test can exclude voxels whose label hangs on rounding
    prob = torch.softmax(res["pred_logits"][0], -1)
    scores, labels = prob.max(-1)
    keep = labels.ne(0) & (scores > 0.3)                       # models/mask3dformer.py:526
    w = (scores[keep].view(-1, 1) * res["pred_masks"][0][keep].sigmoid())
    out["n_kept_queries"] = np.array(int(keep.sum()))
    top2 = torch.topk(w, min(2, w.shape[0]), dim=0).values
    out["label_margin"] = (top2[0] - top2[1]).numpy() if w.shape[0] > 1 else np.ones(masks.shape[1], np.float32)
    info = post["panoptic_seg"][1]
    out["segments"] = np.array([[d["id"], int(d["isthing"]), d["category_id"]] for d in info], np.int64).reshape(-1, 3)
    _save("mask3dformer_at_size", **out)


def scene_fusion_inputs(seed=21, n_vox=24, n_frag=3):
    """fragments for fuse_to_global (direct substitution at the finest scale): coords, tsdf, panoptic ids"""
    rng = np.random.default_rng(seed)
    shifts = [(0, 0, 0), (8, 0, 0), (8, 8, 0)][:n_frag]
    frags = []
    for k, sh in enumerate(shifts):
        occ = rng.random((n_vox,) * 3) < 0.2
        # a solid blob that persists across fragments (in scene coordinates) so that instances re-match
        gx, gy, gz = np.meshgrid(*[np.arange(n_vox)] * 3, indexing="ij")
        sx, sy, sz = gx + sh[0], gy + sh[1], gz + sh[2]
        blob = ((sx - 14) ** 2 + (sy - 10) ** 2 + (sz - 12) ** 2) < 30
        occ |= blob
        xyz = np.argwhere(occ)
        tsdf = np.clip(rng.standard_normal(len(xyz)) * 0.7, -1.2, 1.2).astype(np.float32)[:, None]
        seg = np.zeros(len(xyz), np.int32)
        inblob = blob[xyz[:, 0], xyz[:, 1], xyz[:, 2]]
        seg[inblob] = 1                                     # thing, class 5
        seg[(~inblob) & (xyz[:, 2] < 3)] = 2                # stuff, class 2 (floor)
        seg[(~inblob) & (xyz[:, 2] >= 3) & (rng.random(len(xyz)) < 0.2)] = 3   # another thing, class 7
        info = [{"id": 1, "isthing": True, "category_id": 5}, {"id": 2, "isthing": False, "category_id": 2},
                {"id": 3, "isthing": True, "category_id": 7}]
        frags.append({"coords": np.concatenate([np.zeros((len(xyz), 1), np.int64), xyz], 1).astype(np.int32),
                      "tsdf": tsdf, "seg": seg, "info": info,
                      "origin_partial": (np.array([-0.96, 0.2, -0.4]) + np.array(sh) * 0.04).astype(np.float32)})
    return frags, np.array([-0.96, 0.2, -0.4], np.float32)


def gen_scene_fusion():
    """GRUFusion(direct_substitute=True).forward = NeuralRecon.fuse_to_global (models/gru_fusion.py:259-394
    with panoptic_fusion :133-193 and save_mesh :217-257), three overlapping fragments"""
    from types import SimpleNamespace
    import models.gru_fusion as G

    class PT:
        def __init__(self, F, C):
            self.F, self.C = F, C

        def cuda(self):
            return self

        def detach(self):
            return self

    G.PointTensor = PT
    cfg = SimpleNamespace(THRESHOLDS=[0, 0, 0], VOXEL_SIZE=0.04, N_VOX=[24, 24, 24], N_LAYER=3,
                          FUSION=SimpleNamespace(FULL=True))
    fus = G.GRUFusion(cfg, direct_substitute=True, trianing=False)
    frags, origin = scene_fusion_inputs()
    out = {}
    outputs = {}
    for k, fr in enumerate(frags):
        inputs = {"fragment": ["f"], "scene": ["sceneA"], "vol_origin": torch.from_numpy(origin[None]),
                  "vol_origin_partial": torch.from_numpy(fr["origin_partial"][None])}
        infos = [{"panoptic_seg": [torch.from_numpy(fr["seg"]), [dict(d) for d in fr["info"]]]}]
        outputs = fus(torch.from_numpy(fr["coords"]), torch.from_numpy(fr["tsdf"]), inputs, 2, outputs,
                      save_mesh=(k == len(frags) - 1), panoptic_infos=infos)
        key = f"f{k}_"
        out[key + "map_C"] = fus.global_volume[2].C.numpy().astype(np.int32)
        out[key + "map_F"] = fus.global_volume[2].F.numpy()
        out[key + "instance"] = fus.global_instance.numpy().reshape(-1).astype(np.int32)
        out[key + "semantic"] = fus.global_semantic.numpy().reshape(-1).astype(np.int32)
    out["scene_tsdf"] = outputs["scene_tsdf"][0].numpy()
    out["scene_instance"] = outputs["scene_instance"][0].numpy().astype(np.int32)
    out["scene_semantic"] = outputs["scene_semantic"][0].numpy().astype(np.int32)
    out["scene_origin"] = outputs["origin"][0].numpy()
    _save("scene_fusion", **out)


# ------------------------------------------------------------------------------------------------
# round-2 pins: view mean / variance (G3), aligned-camera coordinates (a7), feat_fusion_pre (G6)
# ------------------------------------------------------------------------------------------------
def _ref_lines(path, first, last):
    """the reference's own source lines between the line containing `first` and the next line
    containing `last` (inclusive), dedented — executed in place of a function the reference does
    not factor out.  Read at generation time only; nothing is copied into this repository."""
    import textwrap
    src = open(os.path.join(ref_shim.REF, path)).read().splitlines()
    i0 = next(i for i, l in enumerate(src) if first in l)
    i1 = next(i for i, l in enumerate(src) if i >= i0 and l.split("#")[0].strip() == last)
    return textwrap.dedent("\n".join(src[i0:i1 + 1])), (i0 + 1, i1 + 1)


def _occ_init_shell(dense=False):
    """A reference Occupancy_Initialization WITHOUT its spconv layers: the constructor cannot run here
    (models/modules.py:257 initialises a spconv weight), so the object is allocated bare and given
    only the dense sub-modules that are importable."""
    import models.occupancy_initialization as M
    import cases
    obj = M.Occupancy_Initialization.__new__(M.Occupancy_Initialization)
    torch.nn.Module.__init__(obj)
    if dense:
        from models.modules import Conv2d_Block, Conv2d_Residual_Block, Fusion_Block
        ch, d = cases.FUSION_PRE_CH, cases.FUSION_PRE_DOWN
        obj.self_fusion_1x, obj.self_fusion_2x, obj.self_fusion_4x = (Fusion_Block(c) for c in ch)
        obj.pool4x = torch.nn.AvgPool2d(2)
        obj.fusion_down = Conv2d_Block(sum(ch), d, 1)
        for i in (1, 2, 3, 4):
            setattr(obj, f"post_fusion_{i}", Conv2d_Residual_Block(d, 3))
    return obj


def gen_occ_init():
    """(G3) Occupancy_Initialization.forward (models/occupancy_initialization.py:61-135) run up to the
    first spconv-dependent line on seeded fused maps: visible-view counts, valid set, per-voxel view
    MEAN and population VARIANCE of the sampled 32-channel features.  `feat_fusion_pre` is replaced
    by the seeded maps; `norm0` captures the variance volume and stops the forward.
    (G6) the reference's own feat_fusion_pre (:41-58) as a whole on seeded weights."""
    import cases

    class _Stop(Exception):
        pass

    out = {}
    for name in cases.OCC_INIT_CASES:
        window, coords, origin, fused, kr = cases.occ_init_case(name)
        obj = _occ_init_shell()
        grabbed = {}

        def norm0(var, grabbed=grabbed):
            grabbed["var"] = var.clone()
            raise _Stop

        obj.norm0 = norm0
        tf = torch.from_numpy(fused)
        obj.feat_fusion_pre = lambda a, b, c, tf=tf: tf[:, 0]
        v, _, _, h, w = fused.shape
        # only the SHAPES of the pyramid enter before feat_fusion_pre: [f4, f8, f16] per view
        views = [[torch.zeros(1, 1, 2 * h, 2 * w), torch.zeros(1, 1, h, w), torch.zeros(1, 1, h // 2, w // 2)]
                 for _ in range(v)]
        shape = tuple(n // 2 for n in window["n_vox"])
        try:
            obj.forward(torch.from_numpy(coords), torch.from_numpy(origin), window["voxel_size"], views,
                        torch.from_numpy(kr), shape, 1, 2)
            raise RuntimeError("forward did not reach norm0")
        except _Stop as e:
            tb = e.__traceback__
            while tb.tb_next is not None and tb.tb_frame.f_code.co_name != "forward":
                tb = tb.tb_next
            loc = tb.tb_frame.f_locals
        var, mean = grabbed["var"].numpy(), loc["mean"].numpy()
        count = loc["count"].numpy()
        n_valid = var.shape[0]
        rows = cases.sample_rows(n_valid, 384, 7)
        out[name + "_count"] = count.astype(np.uint8)
        out[name + "_nvalid"] = np.int64(n_valid)
        out[name + "_rows"] = rows
        out[name + "_var_rows"], out[name + "_mean_rows"] = var[rows], mean[rows]
        out[name + "_var_rowsum"] = var.sum(1, dtype=np.float64).astype(np.float32)[::cases.ROW_STRIDE]
        out[name + "_mean_rowsum"] = mean.sum(1, dtype=np.float64).astype(np.float32)[::cases.ROW_STRIDE]
        out[name + "_subm_coord_rows"] = loc["subm_coords"].numpy()[rows].astype(np.int32)
    # G6: the whole 2D fusion stack
    obj = _occ_init_shell(dense=True)
    cases.seeded_state(obj, 2024)
    obj.train()
    f1, f2, f4 = (torch.from_numpy(a) for a in cases.fusion_pre_inputs())
    with torch.no_grad():
        y = obj.feat_fusion_pre(f1, f2, f4)
    out["fusion_pre_out"] = y.numpy()
    out["fusion_pre_keys"] = np.array(sorted(obj.state_dict().keys()))
    _save("occ_init", **out)


def gen_aligned_coords():
    """(a7) the aligned-camera coordinate block of NeuConNet.forward (models/neucon_network.py:387-398)
    and its twin in GRUFusion.forward (models/gru_fusion.py:332-337), executed from the reference's
    own source lines on seeded voxel lists."""
    from types import SimpleNamespace
    import cases

    code_n, span_n = _ref_lines("models/neucon_network.py", "r_coords = up_coords.detach().clone().float()",
                                "r_coords = r_coords[:, [1, 2, 3, 0]]")
    code_g, span_g = _ref_lines("models/gru_fusion.py", "r_coords = updated_coords.detach().clone().float()",
                                "r_coords = r_coords.permute(1, 0).contiguous()")
    out = {"neucon_lines": np.array(span_n), "gru_lines": np.array(span_g)}
    for scale in range(3):
        coords, origin, w2ac, interval = cases.aligned_case(scale)
        inputs = {"vol_origin_partial": torch.from_numpy(origin), "world_to_aligned_camera": torch.from_numpy(w2ac)}
        ns = {"torch": torch, "up_coords": torch.from_numpy(coords), "bs": 2, "inputs": inputs,
              "self": SimpleNamespace(cfg=SimpleNamespace(VOXEL_SIZE=0.04))}
        exec(code_n, ns)
        out[f"s{scale}_r_coords"] = ns["r_coords"].numpy()
        # GRU-fusion form: batch element i, coordinates in units of this scale's voxel
        for i in range(2):
            m = coords[:, 0] == i
            upd = torch.from_numpy(coords[m][:, 1:].astype(np.int64) // interval)
            ns = {"torch": torch, "updated_coords": upd, "voxel_size": 0.04 * interval, "i": i,
                  "origin": inputs["vol_origin_partial"][i], "inputs": inputs}
            exec(code_g, ns)
            out[f"s{scale}_b{i}_gru_r_coords"] = ns["r_coords"].numpy()
    _save("aligned_coords", **out)


def gen_tsdf_fusion():
    """(f3) TSDFVolumeTorch.integrate (tools/tsdf_fusion/fusion.py:440-485,488-577), the CPU path the reference's data
    pipeline runs per sample (datasets/transforms.py:286-297): 9 synthetic depth frames fused at the three levels.
    Stored: world->camera matrices as torch.inverse produced them (inputs of the restatement), the full weight
    volumes, the occupancy volumes of :295-297, sampled TSDF values and z-slab checksums."""
    import cases
    sys.modules["numba"].njit = lambda *a, **k: (lambda f: f)      # decorators of the unused numba path
    from tools.tsdf_fusion.fusion import TSDFVolumeTorch

    window, depths, intr, poses = cases.tsdf_case()
    out = {"world2cam": np.stack([torch.inverse(torch.from_numpy(p).float()).numpy() for p in poses])}
    for lvl in cases.TSDF_LEVELS:
        dims = torch.tensor([n // 2 ** lvl for n in window["n_vox"]])
        vol = TSDFVolumeTorch(dims, torch.from_numpy(window["vol_origin_partial"]), voxel_size=0.04 * 2 ** lvl, margin=3)
        for v in range(len(depths)):
            vol.integrate(torch.from_numpy(depths[v]), torch.from_numpy(intr[v]), torch.from_numpy(poses[v]), obs_weight=1.)
        tsdf, weight = (t.numpy() for t in vol.get_volume())
        occ = (tsdf < 0.999) & (tsdf > -0.999) & (weight > 1)
        rows = cases.sample_rows(tsdf.size, 20000, 5 + lvl)
        out[f"l{lvl}_weight"] = weight.astype(np.uint8)
        out[f"l{lvl}_occ"] = np.packbits(occ.reshape(-1))
        out[f"l{lvl}_rows"] = rows
        out[f"l{lvl}_tsdf_rows"] = tsdf.reshape(-1)[rows]
        out[f"l{lvl}_tsdf_slab_sums"] = tsdf.sum(axis=(1, 2), dtype=np.float64).astype(np.float32)
    _save("tsdf_fusion", **out)


def criterion_case(seed=3, q=12, n=900, classes=20):
    """seeded decoder outputs + ScanNet-id targets for the set criterion (shared with the tests via cases.py)"""
    import cases
    return cases.criterion_case(seed, q, n, classes)


def gen_criterion():
    """(f4) SetCriterion + HungarianMatcher (models/criterion.py:85-296, models/matcher.py:51-147) and NeuConNet's
    static loss helpers (models/neucon_network.py:627-700) on seeded inputs"""
    import copy
    import cases
    sys.modules["loguru"].logger = type("L", (), {"warning": staticmethod(lambda *a, **k: None)})()
    from models.criterion import SetCriterion
    from models.matcher import HungarianMatcher
    from models.neucon_network import NeuConNet

    outputs, targets = cases.criterion_case()
    to_t = lambda d: {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else [to_t(a) for a in v]) for k, v in d.items()}
    out_t, tgt_t = to_t(outputs), [to_t(t) for t in targets]
    matcher = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0)
    crit = SetCriterion(20, matcher, {}, 0.1, ["labels", "masks"])
    losses = crit(copy.deepcopy(out_t), copy.deepcopy(tgt_t))
    out = {"loss_names": np.array(sorted(losses)), "loss_values": np.array([float(losses[k]) for k in sorted(losses)], np.float64)}
    c = {k: torch.from_numpy(v) for k, v in cases.loss_case().items()}
    out["compute_loss"] = np.float64(NeuConNet.compute_loss(c["tsdf"], c["occ"], c["tsdf_target"], c["occ_target"], mask=c["mask"], pos_weight=1.5))
    out["compute_loss_init"] = np.float64(NeuConNet.compute_loss_init(None, c["occ_init"], c["tsdf_init_target"], c["occ_init_target"]))
    # target look-ups (models/neucon_network.py:117-191) on a bare instance: they read nothing from self
    coords, vols = cases.target_case()
    scale, bs = 1, 2
    lists = lambda v: [None, torch.from_numpy(v), None]
    tin = {"tsdf_list": lists(vols["tsdf"]), "occ_list": lists(vols["occ"]), "semantic_list": lists(vols["semantic"]),
           "instance_list": lists(vols["instance"])}
    bare = NeuConNet.__new__(NeuConNet)
    ct = torch.from_numpy(coords)
    t, o = NeuConNet.get_target(bare, ct, tin, scale)
    ti, oi = NeuConNet.get_target_init(bare, ct, tin, scale)
    out.update(target_tsdf=t.numpy(), target_occ=o.numpy(), target_init_tsdf=ti.numpy(), target_init_occ=oi.numpy())
    for b, tgt in enumerate(NeuConNet.get_panoptic_targets(bare, ct, tin, scale, bs)):
        out[f"panoptic_labels_{b}"] = tgt["labels"].numpy()
        out[f"panoptic_masks_{b}"] = tgt["masks"].numpy()
    # the panoptic-loss tail of NeuConNet.forward, executed from the reference's own lines
    pc = cases.panoptic_loss_case()
    code, span = _ref_lines("models/neucon_network.py", "occ_target_occupancy = occ_target[occupancy].view(-1)",
                            "loss_dict.update({f'panoptic_loss': panoptic_losses})")
    weight_dict = {"loss_ce": 0.2, "loss_mask": 0.8, "loss_dice": 0.8}
    weight_dict.update({f"{k}_{i}": v for i in range(6) for k, v in (("loss_ce", 0.2), ("loss_mask", 0.8), ("loss_dice", 0.8))})
    torch.nn.Module.__init__(bare)
    bare.criterion = SetCriterion(20, HungarianMatcher(cost_class=0.2, cost_mask=0.8, cost_dice=0.8), weight_dict, 0.1, ["labels", "masks"])
    ns = {"torch": torch, "np": np, "self": bare, "bs": 1, "scale": 0, "loss_dict": {}, "panoptic_losses": [], "panoptic_predictions": [],
          "logger": sys.modules["loguru"].logger, "occ_target": torch.from_numpy(pc["occ_target"]),
          "occupancy": torch.from_numpy(pc["occupancy"]), "panoptic_coords": [None, None, torch.from_numpy(pc["coords_fine"])],
          "panoptic_outs": [to_t(pc["outs"])],
          "inputs": {"rgb_list": True, "semantic_list": [torch.from_numpy(pc["semantic"])], "instance_list": [torch.from_numpy(pc["instance"])]}}
    exec(code, ns)
    out["panoptic_loss"] = np.float64(ns["loss_dict"]["panoptic_loss"])
    out["panoptic_loss_lines"] = np.array(span)
    _save("criterion", **out)



def gen_sparse_layers():
    """THE PIN THE BUILD COULD NOT PRODUCE: per-layer activations of the reference's own torchsparse / spconv
    layers — SPVCNN (models/modules.py:75-175 with ops/torchsparse_utils.py:15-105), ConvGRU / SConv3d (:178-222),
    SparseSubMConv3d (:249-271) — with seeded weights on seeded inputs.  Runs only where `torchsparse` and `spconv` import
    (they need their CUDA extensions; neither is installed nor installable in the build container): otherwise it says so and
    writes nothing, and tests/test_sparse_layers_pin.py skips with the same message.  Where it runs it needs a CUDA device
    (`.cuda()` is NOT patched away for this generator: the two libraries have no CPU kernels for these ops)."""
    if not ref_shim.have_real("torchsparse", "torchsparse.nn", "torchsparse.nn.functional", "spconv", "spconv.pytorch"):
        print("sparse_layers: skipped — torchsparse / spconv are not importable here (parity of the sparse layers stays unpinned)")
        return False
    if not torch.cuda.is_available():
        print("sparse_layers: skipped — torchsparse / spconv import, but their kernels need a CUDA device")
        return False
    dev = torch.device("cuda")
    import models.modules as M
    from torchsparse.tensor import PointTensor
    from cases import sparse_layer_inputs
    inp = sparse_layer_inputs()
    out = {"seed": np.array(31)}
    t = lambda a: torch.from_numpy(a).to(dev)

    def record(module, prefix):
        """forward hooks on every leaf-ish submodule: features (and coordinates where the output carries them)"""
        recs, hooks = {}, []

        def hook(name):
            def fn(mod, args, res):
                f = getattr(res, "F", res)
                if torch.is_tensor(f):
                    recs[f"{prefix}/{name}/F"] = f.detach().cpu().numpy()
                    c = getattr(res, "C", None)
                    if torch.is_tensor(c):
                        recs[f"{prefix}/{name}/C"] = c.detach().cpu().numpy()
            return fn
        for name, mod in module.named_modules():
            if name and len(list(mod.children())) == 0:
                hooks.append(mod.register_forward_hook(hook(name)))
        return recs, hooks

    # --- SPVCNN at the finest level's shape (cr = 1/4 -> channels 8, 16, 32, 24, 24; vres = VOXEL_SIZE) ---
    torch.manual_seed(5)
    net = M.SPVCNN(num_classes=1, in_channels=inp["feats"].shape[1], pres=1, cr=0.25, vres=0.04, dropout=False).to(dev).train()
    out.update({f"spvcnn/sd/{k}": v.detach().cpu().numpy() for k, v in net.state_dict().items() if "num_batches" not in k})
    recs, hooks = record(net, "spvcnn")
    with torch.no_grad():
        y = net(PointTensor(t(inp["feats"]), t(inp["pts"])))
    for h_ in hooks:
        h_.remove()
    out.update(recs)
    out["spvcnn/out"] = y.detach().cpu().numpy()

    # --- ConvGRU (both cells of a level share this shape: hidden 12, input 12) ---
    torch.manual_seed(6)
    gru = M.ConvGRU(hidden_dim=12, input_dim=12, pres=1, vres=0.04).to(dev).train()
    out.update({f"convgru/sd/{k}": v.detach().cpu().numpy() for k, v in gru.state_dict().items()})
    recs, hooks = record(gru, "convgru")
    with torch.no_grad():
        hn = gru(PointTensor(t(inp["h"]), t(inp["pts"])), PointTensor(t(inp["x"]), t(inp["pts"])))
    for h_ in hooks:
        h_.remove()
    out.update(recs)
    out["convgru/out"] = hn.detach().cpu().numpy()

    # --- SparseSubMConv3d 3x3x3 and 1x1x1 (spconv weight layout travels with the fixture) ---
    torch.manual_seed(7)
    for k in (3, 1):
        conv = M.SparseSubMConv3d(16, 8, k).to(dev)
        with torch.no_grad():
            conv.sparsesubmconv3d.bias.uniform_(-0.1, 0.1)
            yk = conv(t(inp["sub_feats"]), t(inp["sub_coords"]), [24, 24, 24], 1)
        out[f"subm{k}/weight"] = conv.sparsesubmconv3d.weight.detach().cpu().numpy()
        out[f"subm{k}/bias"] = conv.sparsesubmconv3d.bias.detach().cpu().numpy()
        out[f"subm{k}/out"] = yk.detach().cpu().numpy()
    import torchsparse, spconv  # noqa: E401
    out["torchsparse_version"] = np.array(getattr(torchsparse, "__version__", "?"))
    out["spconv_version"] = np.array(getattr(spconv, "__version__", "?"))
    _save("sparse_layers", **out)
    return True

GENERATORS = {"back_project": gen_back_project, "grid_ops": gen_grid_ops, "dense_blocks": gen_dense_blocks,
              "gru_fusion": gen_gru_fusion, "mask3dformer": gen_mask3dformer, "mask3dformer_at_size": gen_mask3dformer_at_size, "scene_fusion": gen_scene_fusion,
              "occ_init": gen_occ_init, "aligned_coords": gen_aligned_coords, "tsdf_fusion": gen_tsdf_fusion, "criterion": gen_criterion,
              "sparse_layers": gen_sparse_layers}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
