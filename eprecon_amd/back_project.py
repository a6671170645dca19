"""Multi-view back-projection — host-side mirror of the reference operators, running on
libeprecon_hip.so (csrc/back_project.hip).

  back_project(...)        same signature / return as ops/back_project.py:5-80
  Back_Project.forward     same signature / return as models/occupancy_initialization.py:189-261
  view_variance(...)       the sampling + mean/variance block of
                           Occupancy_Initialization.forward (models/occupancy_initialization.py:79-128)

"Nothing to do" is signalled by returning None exactly where the reference does.  Device memory,
streams and the host sync on n_valid are PyTorch's; everything else is the HIP library.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib

MODE_MEAN, MODE_MEAN_DEPTH, MODE_VARIANCE = 0, 1, 2
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
def mark_dense(coords, dims, interval, batch=1):
    """Tag `coords` (int32[B*Dx*Dy*Dz, 4]) as the dense x-major raster of a dims grid at spacing `interval` — what
    generate_grids.dense_coords / ops/generate_grids.py:3-10 produce.  A plain attribute (lost by any op that makes a new
    tensor); nothing in the library branches on it since the brick kernel of round 2 was removed."""
    coords._eprecon_dense = (tuple(int(d) for d in dims), int(interval), int(batch))
    return coords


def _prep_feats(feats):
    """feats: [V, B, C, H, W] (logical shape).  Channels-last storage (stride of C == 1) is passed
    through without a copy; anything else is made NCHW-contiguous."""
    v, b, c, h, w = feats.shape
    if feats.dtype != torch.float32:
        feats = feats.float()
    want = (b * h * w * c, h * w * c, 1, w * c, c)
    if all(sz == 1 or st == ws for sz, st, ws in zip(feats.shape, feats.stride(), want)):
        return feats, LAYOUT_NHWC  # (strides of size-1 dimensions are irrelevant)
    return feats.contiguous(), LAYOUT_NCHW


class PendingBackProject:
    """A back-projection whose kernels are queued on a stream; `result()` waits for the valid counts
    (pinned host copy + event) and returns what `run()` returns.  Lets independent levels be issued
    back to back without a host round trip between them."""

    def __init__(self, tensors, n, v, c, batch, min_valid_per_batch, read, want_grid, want_mean):
        self._t, self._n, self._v, self._c, self._batch = tensors, n, v, c, batch
        self._min_valid, self._read = min_valid_per_batch, read
        self._want_grid, self._want_mean = want_grid, want_mean

    def result(self):
        return self.result_from(self._read.result())     # (deferred checks pending at queue time rode along: _lib.PinnedRead)

    def result_from(self, counts):
        """counts: the 1 + B values of `n_valid_dev` as read by the caller (hold_read: together with other counts)"""
        if any(x < self._min_valid for x in counts[1:]):
            return None  # reference: `return None`
        nv, t, v = counts[0], self._t, self._v
        res = {"feats": t["feats"][:nv], "coords": t["coords"][:nv], "count": t["count"], "n_valid": nv,
               "n_valid_per_batch": counts[1:]}
        if self._want_grid:
            res["grid"] = t["grid"][: v * nv * 2].view(v, nv, 2)
            res["mask"] = t["mask"][: v * nv].view(v, nv).bool()
        if self._want_mean:
            res["mean"] = t["mean"][:nv]
        return res


def run_async(coords, origin, voxel_size, feats, krcam, min_view, mode=MODE_MEAN, min_valid_per_batch=1,
              want_grid=False, want_mean=False, hold_read=False, extra_words=0):
    """Queue the back-projection on the current stream and return a PendingBackProject.
    hold_read: no host copy of the counts is queued — the caller reads `.n_valid_dev` itself, together with whatever else it
    queued on the device count `.n_valid_dev[0:1]` of the compacted rows `.coords_all` (torchsparse_utils.SpvcnnPrefetch), and
    hands the values to result_from().  extra_words: `.n_valid_dev` gets that many more int32 behind the 1 + B counts (the caller's
    own counts, so that one tensor is read)."""
    lib = _lib.load()
    dev = feats.device
    if dev.type != "cuda":
        raise _lib.EpreconError("eprecon_amd operators need device tensors (no CPU fallback)")
    v, b, c, h, w = feats.shape
    n = coords.shape[0]
    coords_i = coords if coords.dtype == torch.int32 else coords.to(torch.int32)
    coords_i = coords_i.contiguous()
    origin_f = origin.to(device=dev, dtype=torch.float32).reshape(-1, 3).contiguous()
    krcam_f = krcam.to(device=dev, dtype=torch.float32).contiguous()
    assert krcam_f.shape == (v, b, 4, 4) and origin_f.shape[0] == b
    feats_c, layout = _prep_feats(feats)
    cout = c + 1 if mode == MODE_MEAN_DEPTH else c

    t = {"feats": torch.empty((n, cout), dtype=torch.float32, device=dev),
         "coords": torch.empty((n, 4), dtype=torch.int32, device=dev),
         "count": torch.empty((n,), dtype=torch.float32, device=dev),
         "mean": torch.empty((n, c), dtype=torch.float32, device=dev) if want_mean else None,
         "grid": torch.empty((v * n * 2,), dtype=torch.float32, device=dev) if want_grid else None,
         "mask": torch.empty((v * n,), dtype=torch.uint8, device=dev) if want_grid else None}
    n_valid_dev = torch.empty((1 + b + int(extra_words),), dtype=torch.int32, device=dev)
    ws_bytes = lib.eprecon_back_project_workspace_bytes(n, b, v, c, h, w, layout)
    ws = _lib.workspace(ws_bytes, dev)
    rc = lib.eprecon_back_project_async(
        _lib.ptr(coords_i), n, _lib.ptr(origin_f), b, float(voxel_size), _lib.ptr(feats_c), layout,
        _lib.ptr(krcam_f), v, c, h, w, int(min_view), mode, _lib.ptr(t["feats"]), _lib.ptr(t["mean"]),
        _lib.ptr(t["coords"]), _lib.ptr(t["count"]), _lib.ptr(t["grid"]), _lib.ptr(t["mask"]),
        _lib.ptr(n_valid_dev), _lib.ptr(ws), ws.numel(), _lib.current_stream())
    _lib.check(rc, "eprecon_back_project_async")
    read = None if hold_read else _lib.PinnedRead(n_valid_dev[:1 + b])
    # inputs stay referenced until result(): the kernels may still be reading them
    t["_keep"] = (coords_i, origin_f, krcam_f, feats_c, n_valid_dev)
    pend = PendingBackProject(t, n, v, c, b, int(min_valid_per_batch), read, want_grid, want_mean)
    pend.n_valid_dev, pend.coords_all = n_valid_dev, t["coords"]
    return pend


def run(coords, origin, voxel_size, feats, krcam, min_view, mode=MODE_MEAN, min_valid_per_batch=1,
        want_grid=False, want_mean=False):
    """Low-level entry: returns None (reference: `return None`) or a dict of device tensors
    {feats [n_valid, C(+1)], coords int32 [n_valid, 4], count f32 [N], n_valid, (grid, mask, mean)}."""
    lib = _lib.load()
    dev = feats.device
    if dev.type != "cuda":
        raise _lib.EpreconError("eprecon_amd operators need device tensors (no CPU fallback)")
    v, b, c, h, w = feats.shape
    n = coords.shape[0]
    coords_i = coords if coords.dtype == torch.int32 else coords.to(torch.int32)
    coords_i = coords_i.contiguous()
    origin_f = origin.to(device=dev, dtype=torch.float32).reshape(-1, 3).contiguous()
    krcam_f = krcam.to(device=dev, dtype=torch.float32).contiguous()
    assert krcam_f.shape == (v, b, 4, 4) and origin_f.shape[0] == b
    feats_c, layout = _prep_feats(feats)
    cout = c + 1 if mode == MODE_MEAN_DEPTH else c

    out_feats = torch.empty((n, cout), dtype=torch.float32, device=dev)
    out_coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
    count = torch.empty((n,), dtype=torch.float32, device=dev)
    out_mean = torch.empty((n, c), dtype=torch.float32, device=dev) if want_mean else None
    out_grid = torch.empty((v * n * 2,), dtype=torch.float32, device=dev) if want_grid else None
    out_mask = torch.empty((v * n,), dtype=torch.uint8, device=dev) if want_grid else None
    n_valid_dev = torch.empty((1 + b,), dtype=torch.int32, device=dev)
    n_valid_host = (ctypes.c_int32 * (1 + b))()
    ws_bytes = lib.eprecon_back_project_workspace_bytes(n, b, v, c, h, w, layout)
    ws = _lib.workspace(ws_bytes, dev)

    if min_view <= 0 and b == 1 and n >= max(int(min_valid_per_batch), 1):
        # every voxel is valid (the view count is never negative; the one batch element owns all rows): nothing to wait for.
        # The count is still produced on the device and checked with the level's next blocking read (a batch index out of
        # range is the one way a row can drop out).
        _lib.check(lib.eprecon_back_project_async(
            _lib.ptr(coords_i), n, _lib.ptr(origin_f), b, float(voxel_size), _lib.ptr(feats_c), layout,
            _lib.ptr(krcam_f), v, c, h, w, int(min_view), mode, _lib.ptr(out_feats), _lib.ptr(out_mean), _lib.ptr(out_coords),
            _lib.ptr(count), _lib.ptr(out_grid), _lib.ptr(out_mask), _lib.ptr(n_valid_dev), _lib.ptr(ws), ws.numel(),
            _lib.current_stream()), "eprecon_back_project_async")
        _lib.defer_check(n_valid_dev[0:1], n, "back-projection with min_view <= 0: rows with a batch index out of range")
        res = {"feats": out_feats, "coords": out_coords, "count": count, "n_valid": n, "n_valid_per_batch": [n]}
        if want_grid:
            res["grid"] = out_grid.view(v, n, 2)
            res["mask"] = out_mask.view(v, n).bool()
        if want_mean:
            res["mean"] = out_mean
        return res
    _lib.count_host_read()
    rc = lib.eprecon_back_project(
        _lib.ptr(coords_i), n, _lib.ptr(origin_f), b, float(voxel_size), _lib.ptr(feats_c), layout,
        _lib.ptr(krcam_f), v, c, h, w, int(min_view), mode, int(min_valid_per_batch),
        _lib.ptr(out_feats), _lib.ptr(out_mean), _lib.ptr(out_coords), _lib.ptr(count),
        _lib.ptr(out_grid), _lib.ptr(out_mask), _lib.ptr(n_valid_dev),
        ctypes.cast(n_valid_host, ctypes.c_void_p), _lib.ptr(ws), ws.numel(), _lib.current_stream())
    _lib.drain_deferred()          # (the library copied the counts itself: pending checks are verified here, when there are any)
    if not _lib.check(rc, "eprecon_back_project"):
        return None
    nv = int(n_valid_host[0])
    res = {"feats": out_feats[:nv], "coords": out_coords[:nv], "count": count, "n_valid": nv,
           "n_valid_per_batch": [int(x) for x in n_valid_host[1:]]}
    if want_grid:
        res["grid"] = out_grid[: v * nv * 2].view(v, nv, 2)
        res["mask"] = out_mask[: v * nv].view(v, nv).bool()
    if want_mean:
        res["mean"] = out_mean[:nv]
    return res


def back_project(coords, origin, voxel_size, feats, KRcam, min_view_number):
    """ops/back_project.py:5-80.  Returns [features f32[N_valid, C+1] (last channel = normalised
    mean depth), coords float32[N_valid, 4], count f32[N]] or None when a batch has no valid voxel."""
    res = run(coords, origin, voxel_size, feats, KRcam, min_view_number, MODE_MEAN_DEPTH)
    if res is None:
        return None
    # the reference concatenates onto torch.empty(0, 4) (float32), so its coords come back as float
    return [res["feats"], res["coords"].to(torch.float32), res["count"]]


class Back_Project(nn.Module):
    """models/occupancy_initialization.py:185-261.  `forward` returns
    [features f32[N_valid, C], coords (input dtype)[N_valid, 4], im_grid, mask, count f32[N]].

    The reference's only caller consumes entries 0, 1 and 4 (models/neucon_network.py:374-378);
    materialising im_grid f32[V, N_valid, 2] and mask bool[V, N_valid] costs 9 extra bytes per
    (view, voxel), so they are produced only when `return_projection` is True, else None."""

    def __init__(self, dim, return_projection=False):
        super().__init__()
        self.return_projection = return_projection

    def forward(self, coords, origin, voxel_size, feats, KRcam, min_view_number):
        if torch.is_grad_enabled() and feats.requires_grad:
            from . import autograd as AG    # training: the gathered features carry the gradient to the image maps
            res = AG.back_project(coords, origin, voxel_size, feats, KRcam, min_view_number, MODE_MEAN)
        else:
            res = run(coords, origin, voxel_size, feats, KRcam, min_view_number, MODE_MEAN,
                      want_grid=self.return_projection)
        if res is None:
            return None
        out_coords = res["coords"] if coords.dtype == torch.int32 else res["coords"].to(coords.dtype)
        return [res["feats"], out_coords, res.get("grid"), res.get("mask"), res["count"]]


def get_img_feats(coords, origin, voxel_size, feats, KRcam, min_view_number):
    """models/occupancy_initialization.py:264-323 (imported by models/neucon_network.py:20, never called there): the view-mean
    features f32[N_valid, C] of the voxels seen by at least `min_view_number` views — entry 0 of Back_Project.forward; an
    empty [0, C] tensor where the reference's loop would have concatenated nothing."""
    res = run(coords, origin, voxel_size, feats, KRcam, min_view_number, MODE_MEAN, min_valid_per_batch=0)
    if res is None:
        return torch.empty((0, feats.shape[2]), dtype=feats.dtype, device=feats.device)
    return res["feats"]


def forward_behind(module, coords, origin, voxel_size, feats, KRcam, min_view_number, behind):
    """Back_Project.forward with more work queued on the DEVICE count of its valid rows in front of the one host read:
    behind(valid_coords int32[N,4] (first n_valid rows live), n_valid_dev int32[1], extra_out int32[behind.n_extra]) ->
    finish(n_valid, host_extra).  -> (what forward returns, finish(...)'s result | None).  Inference only."""
    pend = run_async(coords, origin, voxel_size, feats, KRcam, min_view_number, MODE_MEAN, want_grid=module.return_projection,
                     hold_read=True, extra_words=behind.n_extra)
    nb = pend.n_valid_dev.numel() - behind.n_extra
    finish = behind(pend.coords_all, pend.n_valid_dev[0:1], pend.n_valid_dev[nb:])
    host = _lib.read_counts(pend.n_valid_dev)
    res = pend.result_from(host[:nb])
    if res is None:
        return None, None
    out_coords = res["coords"] if coords.dtype == torch.int32 else res["coords"].to(coords.dtype)
    return [res["feats"], out_coords, res.get("grid"), res.get("mask"), res["count"]], finish(res["n_valid"], host[nb:])


def view_variance(coords, origin, voxel_size, feats_fused, KRcam, min_view_number, min_valid=1000):
    """Per-voxel population variance over the visible views of the fused 32-channel maps
    (models/occupancy_initialization.py:79-128).  Returns None when fewer than `min_valid` voxels
    are valid (:107-108), else dict(var, mean, coords, count, n_valid)."""
    if torch.is_grad_enabled() and feats_fused.requires_grad:
        from . import autograd as AG
        res = AG.back_project(coords, origin, voxel_size, feats_fused, KRcam, min_view_number, MODE_VARIANCE,
                              min_valid_per_batch=min_valid, want_mean=True)
    else:
        res = run(coords, origin, voxel_size, feats_fused, KRcam, min_view_number, MODE_VARIANCE,
                  min_valid_per_batch=min_valid, want_mean=True)
    if res is None:
        return None
    res["var"] = res.pop("feats")
    return res


def to_channels_last(feats):
    """[V, B, C, H, W] -> same logical tensor stored [V, B, H, W, C] with the HIP re-layout kernel,
    for callers that back-project the same maps more than once."""
    lib = _lib.load()
    v, b, c, h, w = feats.shape
    src = feats.float().contiguous()
    dst = torch.empty((v, b, h, w, c), dtype=torch.float32, device=feats.device)
    _lib.check(lib.eprecon_nchw_to_nhwc_async(_lib.ptr(src), _lib.ptr(dst), v * b, c, h * w,
                                               _lib.current_stream()), "eprecon_nchw_to_nhwc_async")
    return dst.permute(0, 1, 4, 2, 3)
