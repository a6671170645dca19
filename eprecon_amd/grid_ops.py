"""Host wrappers for the dense-grid helpers (csrc/grid_ops.hip)."""
import torch

from . import _lib


def init_select(logit, coords, batch_size, dim=24, cell=4, threshold=0.3, must_be_zero=()):
    """models/neucon_network.py:264,298-318.  logit f32[N(,1)] and coords int32[N,4] of the valid
    48^3 voxels -> int32[M,4] stage-0 coordinates (raster order per batch element) and the
    per-batch counts (one host sync, like the reference's torch.nonzero).
    must_be_zero: int32 device scalars ([1]-shaped) read back in the SAME host read; a non-zero one raises (the
    off-grid counters of the dense-grid convolution maps that produced `logit`, sparse.DenseMap)."""
    lib = _lib.load()
    logit = logit.reshape(-1).contiguous()
    coords = coords.contiguous()
    assert coords.dtype == torch.int32 and logit.dtype == torch.float32
    dev = coords.device
    out = torch.empty((batch_size * dim ** 3, 4), dtype=torch.int32, device=dev)
    counts = torch.empty(1 + batch_size, dtype=torch.int32, device=dev)      # (init_select_kernel writes every word)
    ws = _lib.workspace(lib.eprecon_init_select_workspace_bytes(batch_size, dim), dev)
    _lib.check(lib.eprecon_init_select_async(_lib.ptr(logit), _lib.ptr(coords), coords.shape[0],
                                             float(threshold), batch_size, dim, cell, _lib.ptr(out),
                                             _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "eprecon_init_select_async")
    if must_be_zero:
        host = _lib.read_counts(torch.cat([counts] + [t.reshape(1) for t in must_be_zero]))
        if any(host[1 + batch_size:]):
            raise _lib.EpreconError(f"dense-grid convolution: {host[1 + batch_size:]} voxels of the set are not on the grid "
                                    "their VoxelSet was declared with (rows left unwritten); EPRECON_ERR_ARG")
        host = host[:1 + batch_size]
    else:
        host = _lib.read_counts(counts)     # (the off-grid counters of dense-grid maps ride along: sparse.DenseMap defers them)
    return out[: host[0]], host[1:]


class PendingSelect:
    """init_select queued on a stream; result() waits for the counts (pinned host copy + event) and returns what
    init_select returns"""

    def __init__(self, out, read):
        self._out, self._read = out, read

    def result(self):
        host = self._read.result()       # (the deferred checks pending when the selection was queued rode along: _lib.PinnedRead)
        return self._out[: host[0]], host[1:]


def init_select_async(logit, coords, batch_size, dim=24, cell=4, threshold=0.3):
    """init_select without the host round trip: -> PendingSelect"""
    lib = _lib.load()
    logit = logit.reshape(-1).contiguous()
    coords = coords.contiguous()
    assert coords.dtype == torch.int32 and logit.dtype == torch.float32
    dev = coords.device
    out = torch.empty((batch_size * dim ** 3, 4), dtype=torch.int32, device=dev)
    counts = torch.empty(1 + batch_size, dtype=torch.int32, device=dev)      # (init_select_kernel writes every word)
    ws = _lib.workspace(lib.eprecon_init_select_workspace_bytes(batch_size, dim), dev)
    _lib.check(lib.eprecon_init_select_async(_lib.ptr(logit), _lib.ptr(coords), coords.shape[0],
                                             float(threshold), batch_size, dim, cell, _lib.ptr(out),
                                             _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "eprecon_init_select_async")
    return PendingSelect(out, _lib.PinnedRead(counts))


def upsample(pre_feat, pre_coords, interval, up_coords=None):
    """models/neucon_network.py:193-214 -> (up_feat f32[8N, C], up_coords int32[8N, 4]).
    up_coords: the children when an earlier call already wrote them (torchsparse_utils.SpvcnnPrefetch): features only"""
    lib = _lib.load()
    n = pre_coords.shape[0]
    coords = pre_coords if pre_coords.dtype == torch.int32 else pre_coords.to(torch.int32)
    coords = coords.contiguous()
    feat = pre_feat.contiguous()
    c = feat.shape[1]
    if torch.is_grad_enabled() and feat.requires_grad:
        # training: children are parent-major (row 8 i + k), i.e. each row repeated 8 times; autograd sums them back
        up_coords = torch.empty((8 * n, 4), dtype=torch.int32, device=feat.device)
        _lib.check(lib.eprecon_upsample_async(None, 0, _lib.ptr(coords), n, 0, int(interval), None, _lib.ptr(up_coords),
                                              _lib.current_stream()), "eprecon_upsample_async")
        return feat.repeat_interleave(8, dim=0), up_coords
    up_feat = torch.empty((8 * n, c), dtype=torch.float32, device=feat.device)
    if up_coords is not None:
        assert up_coords.shape == (8 * n, 4) and up_coords.dtype == torch.int32
        _lib.check(lib.eprecon_upsample_async(_lib.ptr(feat), feat.stride(0), _lib.ptr(coords), n, c,
                                              int(interval), _lib.ptr(up_feat), None,
                                              _lib.current_stream()), "eprecon_upsample_async")
        return up_feat, up_coords
    up_coords = torch.empty((8 * n, 4), dtype=torch.int32, device=feat.device)
    _lib.check(lib.eprecon_upsample_async(_lib.ptr(feat), feat.stride(0), _lib.ptr(coords), n, c,
                                          int(interval), _lib.ptr(up_feat), _lib.ptr(up_coords),
                                          _lib.current_stream()), "eprecon_upsample_async")
    return up_feat, up_coords


def sparsify(occ, threshold, target, coords, tsdf, feat_all, c_feat, batch_size, behind=None):
    """models/neucon_network.py:454-507 without its random sub-sampling branch, one call + one host read:
    occ f32[N,1], target bool[N] | None, coords int32[N,4], tsdf f32[N,1], feat_all f32[N,C] ->
    (counts [kept, occupied per batch..., occupied & target per batch...], pre_coords, pre_tsdf [M,1], pre_occ [M,1],
    kept_all [M,C], pre_feat [M, c_feat + 2])
    behind (optional): callable(kept_coords int32[N,4] (first M rows live), n_kept_dev int32[1], extra_out int32[behind.n_extra])
    -> finish(m, host_extra): work queued on the DEVICE count of kept rows in front of the read (the next level's coordinate
    side, torchsparse_utils.SpvcnnPrefetch; the panoptic pruning), whose own counts — written into `extra_out`, the tail of this
    call's counts buffer — ride on this read; the call then returns a 7-tuple whose last element is finish(...)'s result"""
    lib = _lib.load()
    n, c_all = feat_all.shape
    dev = feat_all.device
    assert occ.shape[0] == n and tsdf.shape[0] == n and coords.shape == (n, 4) and coords.dtype == torch.int32
    assert coords.is_contiguous() and feat_all.stride(1) == 1
    tgt = None
    if target is not None:
        tgt = target.reshape(-1).contiguous().view(torch.uint8)
    out_coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
    out_tsdf = torch.empty((n, 1), dtype=torch.float32, device=dev)
    out_occ = torch.empty((n, 1), dtype=torch.float32, device=dev)
    out_all = torch.empty((n, c_all), dtype=torch.float32, device=dev)
    out_feat = torch.empty((n, c_feat + 2), dtype=torch.float32, device=dev)
    n_counts = 1 + 2 * batch_size
    counts = torch.empty(n_counts + (behind.n_extra if behind is not None else 0), dtype=torch.int32, device=dev)
    ws = _lib.workspace(lib.eprecon_sparsify_workspace_bytes(n), dev)
    _lib.check(lib.eprecon_sparsify_async(
        _lib.ptr(occ), occ.stride(0), float(threshold), _lib.ptr(tgt), _lib.ptr(coords), _lib.ptr(tsdf), tsdf.stride(0),
        _lib.ptr(feat_all), feat_all.stride(0), c_all, int(c_feat), n, batch_size, _lib.ptr(out_coords), _lib.ptr(out_tsdf),
        _lib.ptr(out_occ), _lib.ptr(out_all), _lib.ptr(out_feat), _lib.ptr(counts), _lib.ptr(ws), ws.numel(),
        _lib.current_stream()), "eprecon_sparsify_async")
    if behind is not None:
        finish = behind(out_coords, counts[0:1], counts[n_counts:])
        host = _lib.read_counts(counts)
        m = host[0]
        return (host[:n_counts], out_coords[:m], out_tsdf[:m], out_occ[:m], out_all[:m], out_feat[:m],
                finish(m, host[n_counts:]))
    host = _lib.read_counts(counts)     # (deferred checks ride on this read: back-projection without a read of its own)
    m = host[0]
    return host, out_coords[:m], out_tsdf[:m], out_occ[:m], out_all[:m], out_feat[:m]
