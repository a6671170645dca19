"""numpy restatement of the reference's TSDF integration — TEST INFRASTRUCTURE (oracle).

Two variants of the same per-voxel update exist in the reference:
  * `integrate` / TSDFVolumeTorch.integrate (tools/tsdf_fusion/fusion.py:440-485,551-575), the CPU torch path the
    data pipeline runs for every sample (datasets/transforms.py:286-295,375-385)         -> variant "torch"
  * the PyCUDA kernel string (tools/tsdf_fusion/fusion.py:67-142, launch :234-259)         -> variant "cuda"
They differ only in how a voxel reaches its pixel:
    torch: cam = inverse(pose) @ [X, 1];  px = round_half_even(cam.x * fx / cam.z + cx);  valid  z > 0
    cuda : cam = R^T (X - t);             px = roundf(fx * (cam.x / cam.z) + cx);          valid  z >= 0
and then share:  d = depth[py, px]; skip if d == 0 (torch: d <= 0) or d - z < -trunc;
    dist = min(1, (d - z) / trunc);  w' = w + obs;  tsdf' = (tsdf * w + obs * dist) / w'.
PINNED against TSDFVolumeTorch (tests/golden/tsdf_fusion.npz, variant "torch"); the "cuda" variant cannot be run
here (PyCUDA) and is restated from the kernel text: parity unpinned for that variant.
"""
import numpy as np

F32 = np.float32


def voxel_centres(dims, origin, voxel_size):
    """X = origin + voxel_size * (ix, iy, iz), x-major raster (fusion.py:517-526), fp32: multiply then add"""
    ax = [np.arange(d, dtype=np.int64) for d in dims]
    g = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    return (F32(voxel_size) * g.astype(F32) + np.asarray(origin, F32)).astype(F32), g


def _fma_rows(m, pts):
    """rows 0..2 of m[4,4] @ [X,1] as the k-ordered fp32 fma chain (= torch's CPU matmul)"""
    import ctypes  # noqa: F401
    x, y, z = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64), pts[:, 2].astype(np.float64)
    out = []
    for r in range(3):
        a = (F32(m[r, 0]) * pts[:, 0]).astype(F32)                       # rounded product
        a = _fma(pts[:, 1], F32(m[r, 1]), a)
        a = _fma(pts[:, 2], F32(m[r, 2]), a)
        a = _fma(np.ones_like(a), F32(m[r, 3]), a)
        out.append(a)
    return out


def _fma(a, b, c):
    """fp32 fused multiply-add: exact product and sum in float64 (24+24 bits fit), one rounding"""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(F32)


def integrate(tsdf, weight, origin, voxel_size, depth, intr, pose_or_w2c, trunc, obs_weight=1.0, variant="torch"):
    """One frame into tsdf / weight (f32[X,Y,Z], modified in place and returned).
    variant "torch": pose_or_w2c = world->camera f32[4,4] (the reference inverts the pose with torch.inverse);
    variant "cuda":  pose_or_w2c = camera pose f32[4,4] (camera->world), used as R^T (X - t)."""
    dims = tsdf.shape
    pts, g = voxel_centres(dims, origin, voxel_size)
    h, w = depth.shape
    fx, fy, cx, cy = F32(intr[0, 0]), F32(intr[1, 1]), F32(intr[0, 2]), F32(intr[1, 2])
    m = np.asarray(pose_or_w2c, F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        if variant == "torch":
            cx_, cy_, cz_ = _fma_rows(m, pts)
            px = np.rint(((cx_ * fx).astype(F32) / cz_).astype(F32) + cx)
            py = np.rint(((cy_ * fy).astype(F32) / cz_).astype(F32) + cy)
            front = cz_ > 0
        else:
            t = (pts - m[:3, 3]).astype(F32)
            cam = []
            for c in range(3):     # cam_pose[0*4+c]*tx + cam_pose[1*4+c]*ty + cam_pose[2*4+c]*tz, left to right, no fma
                a = (m[0, c] * t[:, 0]).astype(F32)
                a = (a + (m[1, c] * t[:, 1]).astype(F32)).astype(F32)
                a = (a + (m[2, c] * t[:, 2]).astype(F32)).astype(F32)
                cam.append(a)
            cx_, cy_, cz_ = cam
            rnd = lambda v: np.where(v >= 0, np.floor(v + F32(0.5)), np.ceil(v - F32(0.5)))   # roundf: half away from 0
            px = rnd(((fx * (cx_ / cz_).astype(F32)).astype(F32) + cx).astype(F32))
            py = rnd(((fy * (cy_ / cz_).astype(F32)).astype(F32) + cy).astype(F32))
            front = ~(cz_ < 0)
    ok = np.isfinite(px) & np.isfinite(py)
    px = np.where(ok, px, -1).astype(np.int64)
    py = np.where(ok, py, -1).astype(np.int64)
    valid = ok & front & (px >= 0) & (px < w) & (py >= 0) & (py < h)
    d = np.zeros(len(pts), F32)
    d[valid] = depth[py[valid], px[valid]]
    diff = (d - cz_).astype(F32)
    valid &= (d > 0) if variant == "torch" else (d != 0)
    valid &= ~(diff < -F32(trunc))
    dist = np.minimum(F32(1.0), (diff / F32(trunc)).astype(F32))
    tv, wv = tsdf.reshape(-1), weight.reshape(-1)
    w_old = wv[valid]
    w_new = (w_old + F32(obs_weight)).astype(F32)
    tv[valid] = (((w_old * tv[valid]).astype(F32) + (F32(obs_weight) * dist[valid]).astype(F32)).astype(F32) / w_new).astype(F32)
    wv[valid] = w_new
    return tsdf, weight


def fuse_views(dims, origin, voxel_size, depths, intrs, mats, margin=3, variant="torch"):
    """TSDFVolumeTorch(dims, origin, voxel_size, margin) + integrate(...) per view + the occupancy rule of the data
    pipeline: occ = |tsdf| < 0.999 and weight > 1 (datasets/transforms.py:295-297,385-387)"""
    tsdf = np.ones(tuple(dims), F32)
    weight = np.zeros(tuple(dims), F32)
    trunc = margin * float(voxel_size)
    for d, k, m in zip(depths, intrs, mats):
        integrate(tsdf, weight, origin, voxel_size, d, k, m, trunc, 1.0, variant)
    occ = (tsdf < F32(0.999)) & (tsdf > F32(-0.999)) & (weight > 1)
    return tsdf, weight, occ
