"""Seeded input generators shared by tests/golden/make_golden.py (build container, reference
present) and the tests (both boxes, reference absent).  numpy default_rng only, so that the
inputs are identical on every machine; nothing here touches /root/reference."""
import numpy as np

from eprecon_amd import synthetic as S

# name -> make_window kwargs; the occupancy initialiser samples the 1/8-resolution fused maps
# (32 channels) on the dense interval-2 grid of the fragment volume
OCC_INIT_CASES = {
    "cfg1": dict(seed=1, width=320, height=240, n_vox=(32, 32, 32)),   # 16^3 = 4,096 voxels, maps 30x40
    "cfg2": dict(seed=0, width=640, height=480, n_vox=(96, 96, 96)),   # 48^3 = 110,592 voxels, maps 60x80
}
OCC_INIT_CH = 32
ROW_STRIDE = 32


def occ_init_case(name):
    """-> window, coords int32[N,4], origin f32[1,3], fused f32[9,1,32,h,w], krcam f32[9,1,4,4]"""
    wargs = OCC_INIT_CASES[name]
    window = S.make_window(**wargs)
    _, h, w = S.pyramid_shapes(wargs["height"], wargs["width"])[1]
    fused = S.make_features(4000 + wargs["seed"], 9, (OCC_INIT_CH, h, w))
    coords = S.dense_coords(window["n_vox"], 2)
    kr = np.ascontiguousarray(window["proj_matrices"][:, 1][:, None])
    origin = window["vol_origin_partial"][None].copy()
    return window, coords, origin, fused, kr


def sample_rows(n, k, seed):
    rng = np.random.default_rng(seed)
    if n <= k:
        return np.arange(n, dtype=np.int64)
    return np.sort(rng.choice(n, size=k, replace=False)).astype(np.int64)


def aligned_case(scale, n=3000, seed=55):
    """voxels of a two-element batch at one scale (interval 4 / 2 / 1 for scale 0 / 1 / 2), grouped by
    batch index, with per-element fragment origins and world->aligned-camera matrices.
    -> coords int32[2n,4] (b,x,y,z), origin f32[2,3], w2ac f32[2,4,4]"""
    interval = 2 ** (2 - scale)
    rng = np.random.default_rng(seed + scale)
    wins = [S.make_window(seed=0), S.make_window(seed=5, advance=0.32)]
    d = 96 // interval
    rows = []
    for b in range(2):
        flat = np.sort(rng.choice(d ** 3, size=n, replace=False))
        xyz = np.stack(np.unravel_index(flat, (d, d, d)), 1) * interval
        rows.append(np.concatenate([np.full((n, 1), b), xyz], 1))
    coords = np.concatenate(rows).astype(np.int32)
    origin = np.stack([w["vol_origin_partial"] for w in wins]).astype(np.float32)
    w2ac = np.stack([w["world_to_aligned_camera"] for w in wins]).astype(np.float32)
    return coords, origin, w2ac, interval


FUSION_PRE_CH = (80, 40, 24)
FUSION_PRE_DOWN = 32
FUSION_PRE_HW = (6, 8)          # 1/16-level map size; the 1/8 and 1/4 levels are x2 and x4


def fusion_pre_inputs(seed=9, views=9):
    """three pyramid levels [V,80,h,w], [V,40,2h,2w], [V,24,4h,4w]"""
    rng = np.random.default_rng(seed)
    h, w = FUSION_PRE_HW
    return [rng.standard_normal((views, c, h * m, w * m)).astype(np.float32)
            for c, m in zip(FUSION_PRE_CH, (1, 2, 4))]


def seeded_state(module, seed, keys=None):
    """Fills the parameters `keys` (default: all) of a torch module from numpy default_rng(seed), in
    sorted key order (weights ~ N(0, 1/fan_in), vectors ~ 1 + 0.2 N or 0.2 N), so that the
    reference's module and this package's module can be given IDENTICAL weights without storing them."""
    import torch
    rng = np.random.default_rng(seed)
    sd = module.state_dict()
    new = {}
    for k in sorted(sd if keys is None else keys):
        v = sd[k]
        if not v.dtype.is_floating_point or k.endswith(("running_mean", "running_var")):
            new[k] = v.clone()
            continue
        a = rng.standard_normal(tuple(v.shape)).astype(np.float32)
        if v.dim() >= 2:
            a *= 1.0 / np.sqrt(max(int(np.prod(v.shape[1:])), 1))
        elif k.endswith("weight"):
            a = 1.0 + 0.2 * a
        else:
            a = 0.2 * a
        new[k] = torch.from_numpy(a.astype(np.float32))
    module.load_state_dict(new, strict=keys is None)
    return module


TSDF_LEVELS = (0, 1, 2)        # voxel size 0.04 * 2**l on (96 / 2**l)^3 voxels, margin 3 (datasets/transforms.py:286-288)


def tsdf_case(seed=0, views=9):
    """depth images f32[V,480,640] (synthetic scene, seeded holes), intrinsics f32[V,3,3], camera poses f32[V,4,4]
    and the fragment origin of one 640x480 window"""
    window = S.make_window(seed=seed)
    depths = np.stack([S.render_depth(window, v, holes_seed=100 + v) for v in range(views)])
    intr = np.repeat(window["intrinsics"][None], views, 0).astype(np.float32)
    return window, depths, intr, window["poses"].astype(np.float32)


def criterion_case(seed=3, q=12, n=900, classes=20, n_aux=2):
    """decoder outputs for q queries over n voxels (+ n_aux auxiliary heads) and ground-truth masks carrying ScanNet
    ids: evaluated classes, an ignored id (13), a tiny (< 100 voxels) mask"""
    rng = np.random.default_rng(seed)
    head = lambda: {"pred_logits": rng.standard_normal((1, q, classes + 1)).astype(np.float32),
                    "pred_masks": (rng.standard_normal((1, q, n)) * 2).astype(np.float32)}
    outputs = head()
    outputs["aux_outputs"] = [head() for _ in range(n_aux)]
    labels = np.array([3, 13, 39, 5, 1, 24], np.int64)          # 13 is not an evaluated class
    owner = rng.integers(0, len(labels) + 1, n)                   # some voxels belong to no mask
    owner[rng.random(n) < 0.5] = 4
    masks = np.stack([owner == k for k in range(len(labels))])
    small = np.zeros(n, bool)
    small[:40] = True
    masks[3] = small & (owner == 3)                               # class 5 with < 100 voxels: dropped
    return outputs, [{"labels": labels, "masks": masks}]


def loss_case(seed=8, n=5000):
    """inputs of NeuConNet.compute_loss / compute_loss_init"""
    rng = np.random.default_rng(seed)
    return {"tsdf": rng.standard_normal(n).astype(np.float32), "occ": rng.standard_normal(n).astype(np.float32),
            "tsdf_target": np.clip(rng.standard_normal(n) * 0.7, -1, 1).astype(np.float32), "occ_target": rng.random(n) < 0.3,
            "mask": rng.random(n) < 0.9, "occ_init": rng.standard_normal(n).astype(np.float32),
            "tsdf_init_target": ((rng.random(n) < 0.4) & (rng.random(n) < 0.8)).astype(np.float32),
            "occ_init_target": (rng.random(n) < 0.5).astype(np.float32)}


def target_case(seed=21, n=4000, dim=24, scale=1, batch=2):
    """voxel coords (b, x, y, z in finest units, multiples of 2**scale) + ground-truth volumes at that scale for the target
    look-ups of NeuConNet (get_target, get_target_init, get_panoptic_targets)"""
    rng = np.random.default_rng(seed)
    coords = np.concatenate([np.sort(rng.integers(0, batch, (n, 1)), axis=0), rng.integers(0, dim, (n, 3)) * 2 ** scale], 1).astype(np.int32)
    vols = {"tsdf": np.clip(rng.standard_normal((batch, dim, dim, dim)), -1, 1).astype(np.float32),
            "occ": rng.random((batch, dim, dim, dim)) < 0.4,
            "semantic": rng.integers(0, 41, (batch, dim, dim, dim)).astype(np.int64),
            "instance": rng.integers(0, 9, (batch, dim, dim, dim)).astype(np.int64)}
    return coords, vols


def panoptic_loss_case(seed=5, m=2600, n_keep=1800, q=10, dim=24, n_aux=2):
    """the tail of NeuConNet.forward (models/neucon_network.py:589-622): finest-level voxels of one fragment (m candidates,
    n_keep of them occupied), their observed-ground-truth flags, decoder outputs over the occupied voxels, label volumes"""
    rng = np.random.default_rng(seed)
    flat = np.sort(rng.choice(dim ** 3, m, replace=False))
    xyz = np.stack(np.unravel_index(flat, (dim, dim, dim)), 1)
    coords_all = np.concatenate([np.zeros((m, 1), np.int64), xyz], 1).astype(np.int32)
    occupancy = np.zeros(m, bool)
    occupancy[np.sort(rng.choice(m, n_keep, replace=False))] = True
    occ_target = (rng.random((m, 1)) < 0.8)
    head = lambda: {"pred_logits": rng.standard_normal((1, q, 21)).astype(np.float32),
                    "pred_masks": (rng.standard_normal((1, q, n_keep)) * 2).astype(np.float32)}
    outs = head()
    outs["aux_outputs"] = [head() for _ in range(n_aux)]
    # a few big instances with evaluated classes so that masks survive the > 100 voxel filter
    blocks = (xyz[:, 0] // 8) + 3 * (xyz[:, 1] // 12)
    instance = np.zeros((1, dim, dim, dim), np.int64)
    semantic = np.zeros((1, dim, dim, dim), np.int64)
    class_of = np.array([3, 5, 7, 13, 24, 39])
    instance[0, xyz[:, 0], xyz[:, 1], xyz[:, 2]] = blocks + 1
    semantic[0, xyz[:, 0], xyz[:, 1], xyz[:, 2]] = class_of[blocks]
    return {"coords_fine": coords_all[occupancy], "occupancy": occupancy, "occ_target": occ_target, "outs": outs,
            "semantic": semantic, "instance": instance}


def mask3d_inputs_at_size(seed=23, c=48, n2=30000, n1=22000, n0=10000):
    """cfg4-size voxel levels on the 96^3 grid for the mask-transformer decoder: every level-1 / level-0 voxel coincides with a
    finest voxel (distance 0), so the reference's float cdist / argmin (models/mask3dformer.py) has a unique nearest voxel and
    no tie to break by rounding noise.  -> [c0, c1, c2] int64[n,3], features f32[1,c,n] per level, mask features f32[1,c,n2]"""
    rng = np.random.default_rng(seed)
    c1 = rng.permutation(np.argwhere(np.ones((48, 48, 48), bool)))[:n1] * 2            # level 1: even coordinates
    c0 = rng.permutation(np.argwhere(np.ones((24, 24, 24), bool)))[:n0] * 4            # level 0: multiples of 4
    anchors = np.unique(np.concatenate([c1, c0]), axis=0)
    extra = np.unique(rng.integers(0, 96, (3 * n2, 3)), axis=0)
    taken = {tuple(v) for v in anchors}
    extra = np.array([v for v in extra if tuple(v) not in taken])
    extra = rng.permutation(extra)[: max(n2 - len(anchors), 0)]
    c2 = rng.permutation(np.concatenate([anchors, extra]))
    feats = [rng.standard_normal((1, c, len(x))).astype(np.float32) for x in (c0, c1, c2)]
    mask_feat = rng.standard_normal((1, c, len(c2))).astype(np.float32)
    return [c0.astype(np.int64), c1.astype(np.int64), c2.astype(np.int64)], feats, mask_feat


def sparse_layer_inputs(seed=31, n=6000, cin=20, extent=3.0):
    """seeded inputs of the sparse-layer pin (also rebuilt by tests/test_sparse_layers_pin.py): a surface-like point cloud in
    metres (two slabs + noise, rotated: the aligned-camera frame of models/neucon_network.py:387-398 is not axis-aligned),
    point features, and the [h | x] rows of a ConvGRU"""
    rng = np.random.default_rng(seed)
    u = rng.uniform(0, extent, (n, 2)).astype(np.float32)
    z = np.where(rng.random(n) < 0.5, 0.6 + 0.2 * np.sin(2 * u[:, 0]), 1.7 + 0.1 * u[:, 1]).astype(np.float32)
    p = np.concatenate([u, z[:, None]], 1) + rng.normal(0, 0.01, (n, 3)).astype(np.float32)
    a = 0.3
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    pts = np.concatenate([p @ rot.T, np.zeros((n, 1), np.float32)], 1).astype(np.float32)      # (x, y, z, batch) like r_coords
    feats = rng.standard_normal((n, cin)).astype(np.float32)
    h = rng.standard_normal((n, 12)).astype(np.float32) * 0.5
    x = rng.standard_normal((n, 12)).astype(np.float32) * 0.5
    grid = np.unique(rng.integers(0, 24, (4000, 3)), axis=0).astype(np.int32)
    grid = np.concatenate([np.zeros((len(grid), 1), np.int32), grid], 1)                          # (b, x, y, z) like spconv's indices
    sub_f = rng.standard_normal((len(grid), 16)).astype(np.float32)
    return {"pts": pts, "feats": feats, "h": h, "x": x, "sub_coords": grid, "sub_feats": sub_f}
