"""numpy restatement of the small dense-grid steps of NeuConNet.forward (test infrastructure).

Pinned against golden vectors captured from the reference's own generate_grid / upsample /
erode / dilate (tests/golden/make_golden.py -> grid_ops.npz)."""
import numpy as np


def generate_grid(n_vox, interval):
    """ops/generate_grids.py:3-10 -> (f32[3, N] x-major raster, dims)"""
    ax = [np.arange(0, n_vox[a], interval) for a in range(3)]
    g = np.stack(np.meshgrid(ax[0], ax[1], ax[2], indexing="ij")).reshape(3, -1).astype(np.float32)
    return g, tuple(len(a) for a in ax)


def upsample(feat, coords, interval):
    """models/neucon_network.py:193-214 -> (feat[8N, C], coords[8N, 4])"""
    child = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1]])
    up_c = np.repeat(np.asarray(coords)[:, None, :], 8, axis=1).copy()
    up_c[:, :, 1:] += child[None] * interval
    up_f = np.repeat(np.asarray(feat)[:, None, :], 8, axis=1)
    return up_f.reshape(-1, feat.shape[1]), up_c.reshape(-1, 4)


def _box_sum(vol):
    d = vol.shape[0]
    p = np.zeros((d + 2,) * 3, np.int32)
    p[1:-1, 1:-1, 1:-1] = vol
    s = np.zeros((d,) * 3, np.int32)
    for dx in range(3):
        for dy in range(3):
            for dz in range(3):
                s += p[dx:dx + d, dy:dy + d, dz:dz + d]
    return s


def erode(vol):
    return _box_sum(vol.astype(np.int32)) == 27


def dilate(vol):
    return _box_sum(vol.astype(np.int32)) >= 1


def init_select(logit, coords, batch, dim=24, cell=4, thr=0.3):
    """models/neucon_network.py:264,298-318: sigmoid > thr on the valid voxels -> 2^3 OR-pool ->
    erode -> dilate -> dilate -> nonzero (raster) * cell, batch index prepended."""
    logit = np.asarray(logit, np.float32).reshape(-1)
    sig = (np.float32(1.0) / (np.float32(1.0) + np.exp(-logit, dtype=np.float32))).astype(np.float32)
    sel = sig > np.float32(thr)
    out = []
    for b in range(batch):
        m = (coords[:, 0] == b) & sel
        vol = np.zeros((dim,) * 3, bool)
        c = coords[m][:, 1:] // cell
        vol[c[:, 0], c[:, 1], c[:, 2]] = True
        vol = dilate(dilate(erode(vol)))
        xyz = np.argwhere(vol) * cell
        out.append(np.concatenate([np.full((len(xyz), 1), b), xyz], 1))
    return np.concatenate(out, 0).astype(np.int32)
