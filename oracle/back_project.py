"""ctypes front-end of oracle/c/back_project_oracle.c (see that file for the arithmetic contract
and the reference lines it follows)."""
import ctypes

import numpy as np

from . import lib

MODE_MEAN, MODE_MEAN_DEPTH, MODE_VARIANCE = 0, 1, 2


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def back_project(coords, origin, voxel_size, feats, krcam, min_view, mode=MODE_MEAN,
                 want_grid=False):
    """coords int32[N,4] bxyz; origin f32[B,3]; feats f32[V,B,C,H,W]; krcam f32[V,B,4,4].

    Returns None when some batch has no valid voxel (reference: `return None`), else a dict with
    feats f32[n_valid, C(+1)], coords int32[n_valid,4], count f32[N], and optionally
    grid f32[V,n_valid,2], mask bool[V,n_valid], mean f32[n_valid,C] (variance mode)."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    origin = np.ascontiguousarray(origin, dtype=np.float32).reshape(-1, 3)
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    krcam = np.ascontiguousarray(krcam, dtype=np.float32)
    V, B, C, H, W = feats.shape
    assert krcam.shape == (V, B, 4, 4) and origin.shape[0] == B
    n = coords.shape[0]
    cout = C + 1 if mode == MODE_MEAN_DEPTH else C
    out_feats = np.zeros((n, cout), np.float32)
    out_mean = np.zeros((n, C), np.float32) if mode == MODE_VARIANCE else None
    out_coords = np.zeros((n, 4), np.int32)
    count = np.zeros((n,), np.float32)
    fn = lib().eprecon_oracle_back_project
    fn.restype = ctypes.c_int64
    args = lambda grid, mask: (
        _p(coords), ctypes.c_int64(n), _p(origin), ctypes.c_int(B), ctypes.c_float(voxel_size),
        _p(feats), _p(krcam), ctypes.c_int(V), ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W),
        ctypes.c_int(min_view), ctypes.c_int(mode), _p(out_feats), _p(out_mean), _p(out_coords),
        _p(count), _p(grid), _p(mask))
    nv = fn(*args(None, None))
    if nv < 0:
        return None
    res = {"count": count}
    if want_grid:
        grid = np.zeros((V, nv, 2), np.float32)
        mask = np.zeros((V, nv), np.uint8)
        nv2 = fn(*args(grid, mask))
        assert nv2 == nv
        res["grid"] = grid
        res["mask"] = mask.astype(bool)
    res["feats"] = out_feats[:nv].copy()
    res["coords"] = out_coords[:nv].copy()
    if out_mean is not None:
        res["mean"] = out_mean[:nv].copy()
    return res


def num_threads():
    fn = lib().eprecon_oracle_num_threads
    fn.restype = ctypes.c_int
    return int(fn())


def set_threads(n):
    lib().eprecon_oracle_set_threads(ctypes.c_int(int(n)))
