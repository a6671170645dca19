"""Seeded synthetic fragment windows (numpy only, so the build container and the GPU box agree).

Produces the tensors the reference's data pipeline hands to the model boundary
(datasets/transforms.py:41-80 `IntrinsicsPoseToProjection`, :250-260 `vol_origin_partial`):

  proj_matrices          f32[V, 3, 4, 4]   level l uses K / 4 / 2**l with K[2,2] = 1, times inv(pose)
  world_to_aligned_camera f32[4, 4]        gravity-aligned middle-camera frame
  vol_origin, vol_origin_partial f32[3]
  tsdf_list / occ_list   analytic scene (box room + spheres) sampled at 0.04/0.08/0.16 m

No dataset or checkpoint exists in this environment, so every benchmark and parity test runs on
these windows.
"""
import numpy as np

from .config import CH_IMG, N_VIEWS

F32 = np.float32


def _look_at_pose(eye, forward):
    """camera-to-world 4x4 for a camera at `eye` looking along `forward`; camera axes are
    x right, y down, z forward; the world is z-up (ScanNet convention)."""
    f = np.asarray(forward, dtype=np.float64)
    f = f / np.linalg.norm(f)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(f, up)
    right /= np.linalg.norm(right)
    down = np.cross(f, right)
    pose = np.eye(4)
    pose[:3, 0] = right
    pose[:3, 1] = down
    pose[:3, 2] = f
    pose[:3, 3] = eye
    return pose


def _rodrigues(axis, theta):
    axis = axis / np.linalg.norm(axis)
    kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(theta) * kx + (1 - np.cos(theta)) * (kx @ kx)


def world_to_aligned_camera(middle_pose):
    """datasets/transforms.py:48-63: rotate the middle camera so that world-up maps to camera -y."""
    w2c = np.linalg.inv(middle_pose)
    z_c = (w2c @ np.array([0.0, 0.0, 1.0, 0.0]))[:3]
    target = np.array([0.0, -1.0, 0.0])
    axis = np.cross(z_c, target)
    n = np.linalg.norm(axis)
    if n < 1e-12:
        rot = np.eye(3)
    else:
        theta = np.arccos(np.clip(-z_c[1] / np.linalg.norm(z_c), -1.0, 1.0))
        rot = _rodrigues(axis, theta)
    r4 = np.eye(4)
    r4[:3, :3] = rot
    return (r4.astype(F32) @ w2c.astype(F32)).astype(F32)


def intrinsics_for(width, height):
    """ScanNet colour intrinsics rescaled from 1296x968 to the working image size."""
    s = width / 1296.0
    k = np.eye(3)
    k[0, 0] = k[1, 1] = 577.87 * s
    k[0, 2] = (width - 1) / 2.0
    k[1, 2] = (height - 1) / 2.0
    return k


def make_window(seed=0, width=640, height=480, n_views=N_VIEWS, n_vox=(96, 96, 96),
                voxel_size=0.04, advance=0.0, stride=4):
    """One 9-view fragment window. `advance` shifts the camera arc along +x (metres) so that
    consecutive fragments of one scene overlap (GRU-fusion tests)."""
    rng = np.random.default_rng(seed)
    k = intrinsics_for(width, height)
    poses = []
    for v in range(n_views):
        t = v - (n_views - 1) / 2.0
        yaw = np.deg2rad(5.0 * t)                 # 40 degree arc over 9 views
        pitch = np.deg2rad(-12.0 + rng.uniform(-1.0, 1.0))
        fwd = np.array([np.sin(yaw) * np.cos(pitch), np.cos(yaw) * np.cos(pitch), np.sin(pitch)])
        eye = np.array([0.1 * t + advance, -0.6 + rng.uniform(-0.02, 0.02), 1.5 + rng.uniform(-0.02, 0.02)])
        poses.append(_look_at_pose(eye, fwd))
    poses = np.stack(poses)

    proj = np.zeros((n_views, 3, 4, 4), dtype=F32)
    for v in range(n_views):
        w2c = np.linalg.inv(poses[v]).astype(F32)
        for lvl in range(3):
            ks = (k / stride / 2 ** lvl).astype(F32)
            ks[2, 2] = 1.0
            m = w2c.copy()
            m[:3, :4] = ks @ w2c[:3, :4]
            proj[v, lvl] = m

    extent = np.array(n_vox, dtype=np.float64) * voxel_size
    vol_origin = np.array([-extent[0] / 2, 0.2, -0.4])
    # snap the fragment origin to a multiple of 8 finest voxels relative to the scene origin
    shift = np.array([advance, 0.0, 0.0])
    partial = vol_origin + np.round(shift / (8 * voxel_size)) * (8 * voxel_size)
    return {
        "proj_matrices": proj,
        "poses": poses.astype(F32),
        "intrinsics": k.astype(F32),
        "vol_origin": vol_origin.astype(F32),
        "vol_origin_partial": partial.astype(F32),
        "world_to_aligned_camera": world_to_aligned_camera(poses[n_views // 2]),
        "voxel_size": voxel_size,
        "n_vox": tuple(n_vox),
        "image_hw": (height, width),
    }


def pyramid_shapes(height=480, width=640, channels=CH_IMG):
    """(C, H, W) per projection level l = 0 (1/4 res), 1 (1/8), 2 (1/16).  The backbone emits
    [24,120,160] / [40,60,80] / [80,30,40] for 640x480 (models/neuralrecon.py:52)."""
    out = []
    for lvl in range(3):
        c = channels[2 - lvl]
        out.append((c, height // (4 * 2 ** lvl), width // (4 * 2 ** lvl)))
    return out


def make_features(seed, n_views, shape_chw, batch=1):
    """i.i.d. N(0,1) feature maps f32[V, B, C, H, W]."""
    rng = np.random.default_rng(seed)
    c, h, w = shape_chw
    return rng.standard_normal((n_views, batch, c, h, w), dtype=F32)


def dense_coords(n_vox, interval, batch=1):
    """int32[N,4] bxyz of the dense (n/interval)^3 grid in finest-voxel units, x-major raster
    (same order as ops/generate_grids.py:3-10 followed by models/neucon_network.py:247-251)."""
    ax = [np.arange(0, n_vox[a], interval, dtype=np.int32) for a in range(3)]
    gx, gy, gz = np.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
    xyz = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1)
    rows = []
    for b in range(batch):
        rows.append(np.concatenate([np.full((xyz.shape[0], 1), b, np.int32), xyz], axis=1))
    return np.ascontiguousarray(np.concatenate(rows, axis=0))


def analytic_tsdf(window, level, trunc_voxels=3.0):
    """Analytic TSDF of a box room plus three spheres on the (96 / 2**level)^3 grid at voxel size
    0.04 * 2**level, truncated at `trunc_voxels` voxels and normalised to [-1, 1]."""
    vs = window["voxel_size"] * 2 ** level
    dims = [n // 2 ** level for n in window["n_vox"]]
    o = window["vol_origin_partial"].astype(np.float64)
    ax = [o[a] + np.arange(dims[a]) * vs for a in range(3)]
    x, y, z = np.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
    # room: floor z = 0, back wall y = 3.4, side walls x = +-1.7 (signed distance, positive inside)
    d = np.minimum.reduce([z - 0.0, 3.4 - y, x + 1.7, 1.7 - x])
    for cx, cy, cz, r in [(-0.6, 2.0, 0.5, 0.5), (0.7, 2.4, 0.35, 0.35), (0.1, 1.5, 0.25, 0.25)]:
        d = np.minimum(d, np.sqrt((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) - r)
    t = np.clip(d / (trunc_voxels * vs), -1.0, 1.0)
    return t.astype(F32)


SCENE_SPHERES = [(-0.6, 2.0, 0.5, 0.5), (0.7, 2.4, 0.35, 0.35), (0.1, 1.5, 0.25, 0.25)]


def scene_sdf(x, y, z):
    """signed distance (positive in free space) of the analytic scene of analytic_tsdf: floor z = 0, back wall
    y = 3.4, side walls x = +-1.7, three spheres"""
    d = np.minimum.reduce([z - 0.0, 3.4 - y, x + 1.7, 1.7 - x])
    for cx, cy, cz, r in SCENE_SPHERES:
        d = np.minimum(d, np.sqrt((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) - r)
    return d


def analytic_panoptic(window, level=0):
    """(semantic, instance) int64 grids of the analytic scene: each voxel takes the label of the nearest surface
    primitive — floor (ScanNet id 2), the three walls (id 1, one instance each), the three spheres (chair 5, sofa 6,
    table 7).  Instance ids are 1-based and unique per primitive; what datasets/transforms.py hands the criterion
    as inputs['semantic_list'] / ['instance_list']."""
    vs = window["voxel_size"] * 2 ** level
    dims = [n // 2 ** level for n in window["n_vox"]]
    o = window["vol_origin_partial"].astype(np.float64)
    ax = [o[a] + np.arange(dims[a]) * vs for a in range(3)]
    x, y, z = np.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
    prims = [z - 0.0, 3.4 - y, x + 1.7, 1.7 - x]
    prims += [np.sqrt((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2) - r for cx, cy, cz, r in SCENE_SPHERES]
    nearest = np.argmin(np.abs(np.stack(prims)), axis=0)
    semantic_of = np.array([2, 1, 1, 1, 5, 6, 7], np.int64)
    return semantic_of[nearest], (nearest + 1).astype(np.int64)


def render_depth(window, view, max_depth=6.0, holes_seed=None, hole_fraction=0.03):
    """Synthetic depth image f32[H, W] (camera-frame z in metres, 0 = invalid) of the analytic scene seen from
    view `view` of the window, by sphere tracing.  Rays that leave the scene (there is no ceiling / front wall)
    and a seeded random `hole_fraction` of the pixels are invalid — what a real depth sensor hands to the
    TSDF integration (tools/tsdf_fusion/fusion.py of the reference)."""
    h, w = window["image_hw"]
    k = window["intrinsics"].astype(np.float64)
    pose = window["poses"][view].astype(np.float64)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d_cam = np.stack([(u - k[0, 2]) / k[0, 0], (v - k[1, 2]) / k[1, 1], np.ones_like(u)], -1)
    d = d_cam @ pose[:3, :3].T                       # world direction per unit of camera z
    speed = np.linalg.norm(d, axis=-1)
    eye = pose[:3, 3]
    t = np.full((h, w), 0.05)
    for _ in range(80):
        p = eye + t[..., None] * d
        dist = scene_sdf(p[..., 0], p[..., 1], p[..., 2])
        t = np.minimum(t + np.maximum(dist, 0.0) / speed, max_depth + 1.0)
    p = eye + t[..., None] * d
    ok = (np.abs(scene_sdf(p[..., 0], p[..., 1], p[..., 2])) < 2e-3) & (t < max_depth)
    depth = np.where(ok, t, 0.0).astype(F32)
    if holes_seed is not None:
        rng = np.random.default_rng(holes_seed)
        depth[rng.random((h, w)) < hole_fraction] = 0.0
    return depth


def make_model_inputs(windows, feat_seed=0, scene="scene0000_00", fragment_ids=None, panoptic=False):
    """numpy inputs of NeuConNet.forward for a batch of windows (list of make_window dicts):
    both backbones' pyramids as the reference's list over views of [f4, f8, f16] (each [B,C,H,W]),
    and the `inputs` dict of datasets/transforms.py (proj_matrices [B,V,3,4,4], origins,
    world_to_aligned_camera, analytic tsdf_list / occ_list at 0.04 / 0.08 / 0.16 m)."""
    b = len(windows)
    h, w = windows[0]["image_hw"]
    shapes = pyramid_shapes(h, w)
    rng = np.random.default_rng(feat_seed)

    def pyramid():
        return [[rng.standard_normal((b,) + shapes[lvl], dtype=F32) for lvl in range(3)] for _ in range(N_VIEWS)]

    features, features_occ_pano = pyramid(), pyramid()
    tsdf_list, occ_list = [], []
    for lvl in range(3):
        t = np.stack([analytic_tsdf(wd, lvl) for wd in windows])
        tsdf_list.append(t)
        occ_list.append(np.abs(t) < 0.999)
    inputs = {
        "proj_matrices": np.stack([wd["proj_matrices"] for wd in windows]),
        "vol_origin": np.stack([wd["vol_origin"] for wd in windows]),
        "vol_origin_partial": np.stack([wd["vol_origin_partial"] for wd in windows]),
        "world_to_aligned_camera": np.stack([wd["world_to_aligned_camera"] for wd in windows]),
        "scene": [scene] * b,
        "fragment": fragment_ids or [f"{scene}_{k}" for k in range(b)],
        "tsdf_list": tsdf_list,
        "occ_list": occ_list,
    }
    if panoptic:   # ground truth of the panoptic criterion (training); 'rgb_list' is the key the reference tests for
        labels = [[analytic_panoptic(wd, lvl) for wd in windows] for lvl in range(3)]
        inputs["semantic_list"] = [np.stack([sem for sem, _ in lv]) for lv in labels]
        inputs["instance_list"] = [np.stack([ins for _, ins in lv]) for lv in labels]
        inputs["rgb_list"] = [np.zeros(t.shape + (3,), np.uint8) for t in tsdf_list]
    return features, features_occ_pano, inputs


def to_device(obj, device):
    """numpy arrays -> torch tensors on `device`, recursively (lists / dicts; strings pass through)"""
    import torch
    if isinstance(obj, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(obj)).to(device)
    if isinstance(obj, dict):
        out = {k: to_device(v, device) for k, v in obj.items()}
        # the volume origins also stay on the host, where the data loader built them (datasets/transforms.py:250-260):
        # GRUFusion reads them there instead of synchronising the device once per fragment
        for k in ("vol_origin", "vol_origin_partial"):
            if isinstance(obj.get(k), np.ndarray):
                out[k + "_host"] = torch.from_numpy(np.ascontiguousarray(obj[k]).astype(np.float32))
        return out
    if isinstance(obj, (list, tuple)):
        return [to_device(v, device) for v in obj]
    return obj
