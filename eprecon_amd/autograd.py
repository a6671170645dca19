"""torch.autograd Functions over libeprecon_hip.so — the training side of the sparse operators.

The reference trains through torchsparse's / spconv's own autograd (main.py:181-313: loss.backward() through
spnn.Conv3d, SubMConv3d, spvoxelize, spdevoxelize and F.grid_sample).  Here every forward is the same HIP entry point
the inference path calls, and every backward is a HIP kernel as well (csrc/backward.hip, csrc/back_project.hip):

  sparse_conv        dx: the forward gather-GEMM on the inverted kernel map with transposed weights;
                     dW: eprecon_sparse_conv_wgrad_async (fp32 MFMA over compacted rows, deterministic); db: column sums
  devoxelize         scatter of the trilinear weights (hardware float atomics)
  segment_mean       gather of the voxel gradient scaled by 1 / count
  back_project       scatter of the bilinear taps into the channels-last feature maps

The functional wrappers fall through to the plain (non-recording) call when no input requires a gradient.
"""
import torch

from . import _lib
from . import back_project as BP
from . import sparse as SP

__all__ = ["sparse_conv", "devoxelize", "segment_mean", "back_project", "inverse_map"]


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def inverse_map(nbr, n_in):
    """inv int32[K, n_in] with inv[k][j] = i <=> nbr[k][i] = j (cached on the map tensor: the kernel maps of a voxel
    set are shared by all the convolutions of a forward pass, and so are their inverses)"""
    cache = getattr(nbr, "_eprecon_inverse", None)
    if cache is None:
        cache = nbr._eprecon_inverse = {}
    inv = cache.get(n_in)
    if inv is None:
        lib = _lib.load()
        inv = torch.empty((nbr.shape[0], n_in), dtype=torch.int32, device=nbr.device)
        _lib.check(lib.eprecon_invert_map_async(_lib.ptr(nbr), nbr.shape[0], nbr.shape[1], n_in, _lib.ptr(inv),
                                                _lib.current_stream()), "eprecon_invert_map_async")
        cache[n_in] = inv
    return inv


def conv_weight_grad(x, dy, nbr, kvol, cin, cout):
    lib = _lib.load()
    n_out = dy.shape[0]
    dw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=x.device)
    ws = torch.empty((max(lib.eprecon_sparse_conv_wgrad_workspace_bytes(kvol, n_out, cin, cout), 16),), dtype=torch.uint8,
                     device=x.device)
    _lib.check(lib.eprecon_sparse_conv_wgrad_async(_lib.ptr(x), x.stride(0), _lib.ptr(dy), dy.stride(0), _lib.ptr(nbr), kvol,
                                                   n_out, cin, cout, _lib.ptr(dw), _lib.ptr(ws), ws.numel(),
                                                   _lib.current_stream()), "eprecon_sparse_conv_wgrad_async")
    return dw


class _SparseConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, nbr):
        w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
        x = x if x.stride(1) == 1 else x.contiguous()
        ctx.save_for_backward(x, w3)
        ctx.nbr, ctx.has_bias, ctx.w_dim = nbr, bias is not None, weight.dim()
        return SP.sparse_conv(x, w3, nbr, bias)

    @staticmethod
    def backward(ctx, dy):
        x, w3 = ctx.saved_tensors
        nbr = ctx.nbr
        dy = dy.contiguous()
        kvol, cin, cout = w3.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = w3.transpose(1, 2).contiguous()
            if nbr is None:
                dx = SP.sparse_conv(dy, wt, None)
            else:
                dx = SP.sparse_conv(dy, wt, inverse_map(nbr, x.shape[0]))
        if ctx.needs_input_grad[1]:
            dw = conv_weight_grad(x, dy, nbr, kvol, cin, cout)
            if ctx.w_dim == 2:
                dw = dw[0]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db, None


def sparse_conv(x, weight, nbr=None, bias=None):
    """y[i] = bias + sum_k x[nbr[k][i]] @ weight[k], differentiable in x, weight and bias"""
    if not _needs_grad(x, weight, bias):
        return SP.sparse_conv(x, weight, nbr, bias)
    return _SparseConv.apply(x, weight, bias, nbr)


def corner_lists(idx8, m):
    """CSR lists voxel -> (point, corner) entries of a corner table idx8 int32[n, 8] (eprecon_segment_lists_async over the
    flattened table; missing corners skipped), cached ON the table tensor (`idx8._csr`): the lists die with the table
    instead of pinning up to 24 tables + lists in a process-global list across training iterations"""
    hit = getattr(idx8, "_csr", None)
    if hit is not None and hit[0] == m:
        return hit[1], hit[2]
    lib = _lib.load()
    flat = idx8.reshape(-1)
    n8, dev = flat.shape[0], idx8.device
    offsets = torch.empty(m + 1, dtype=torch.int32, device=dev)
    order = torch.empty(max(n8, 1), dtype=torch.int32, device=dev)
    ws = _lib.workspace(lib.eprecon_segment_workspace_bytes(n8, m), dev)
    _lib.check(lib.eprecon_segment_lists_async(_lib.ptr(flat), n8, m, _lib.ptr(offsets), _lib.ptr(order), _lib.ptr(ws), ws.numel(),
                                               _lib.current_stream()), "eprecon_segment_lists_async")
    idx8._csr = (m, offsets, order)
    return offsets, order


class _Devoxelize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx8, w8):
        lib = _lib.load()
        feat = feat if feat.stride(1) == 1 else feat.contiguous()
        n, c = idx8.shape[0], feat.shape[1]
        out = torch.empty((n, c), dtype=torch.float32, device=feat.device)
        _lib.check(lib.eprecon_devoxelize_async(_lib.ptr(feat), feat.stride(0), _lib.ptr(idx8), _lib.ptr(w8), n, c, _lib.ptr(out),
                                                out.stride(0), 0, _lib.current_stream()), "eprecon_devoxelize_async")
        ctx.idx8, ctx.w8, ctx.m = idx8, w8, feat.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        dout = dout.contiguous()
        n, c = dout.shape
        dfeat = torch.empty((ctx.m, c), dtype=torch.float32, device=dout.device)
        # run-to-run bit-identical: no float atomics — the entries that reach a voxel are summed in the order of a CSR list that
        # depends on the corner table only (built once per table, shared by the layers that devoxelise with it)
        offsets, order = corner_lists(ctx.idx8, ctx.m)
        _lib.check(lib.eprecon_devoxelize_backward_csr_async(_lib.ptr(dout), dout.stride(0), _lib.ptr(ctx.w8), _lib.ptr(offsets),
                                                             _lib.ptr(order), ctx.m, c, _lib.ptr(dfeat), dfeat.stride(0),
                                                             _lib.current_stream()), "eprecon_devoxelize_backward_csr_async")
        return dfeat, None, None


def devoxelize(feat, idx8, w8):
    """out[p] = sum_c w8[p, c] * feat[idx8[p, c]]  (trilinear voxel -> point transfer)"""
    return _Devoxelize.apply(feat, idx8, w8)


class _SegmentMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx, offsets, order, m):
        lib = _lib.load()
        feat = feat if feat.stride(1) == 1 else feat.contiguous()
        c = feat.shape[1]
        out = torch.empty((m, c), dtype=torch.float32, device=feat.device)
        _lib.check(lib.eprecon_segment_mean_async(_lib.ptr(feat), feat.stride(0), _lib.ptr(offsets), _lib.ptr(order), m, c,
                                                  _lib.ptr(out), out.stride(0), _lib.current_stream()), "eprecon_segment_mean_async")
        ctx.idx, ctx.offsets, ctx.n = idx, offsets, feat.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        dout = dout.contiguous()
        c = dout.shape[1]
        counts = (ctx.offsets[1:] - ctx.offsets[:-1]).to(torch.float32)
        scale = torch.where(counts > 0, 1.0 / counts.clamp(min=1.0), torch.zeros_like(counts))
        dfeat = torch.empty((ctx.n, c), dtype=torch.float32, device=dout.device)
        _lib.check(lib.eprecon_gather_rows_scaled_async(_lib.ptr(dout), dout.stride(0), _lib.ptr(ctx.idx), _lib.ptr(scale), ctx.n, c,
                                                        _lib.ptr(dfeat), dfeat.stride(0), _lib.current_stream()),
                   "eprecon_gather_rows_scaled_async")
        return dfeat, None, None, None, None


def segment_mean(feat, idx, lists, m):
    """mean of the rows of feat per voxel; idx int32[n] point -> voxel (-1: dropped), lists = its CSR form"""
    offsets, order = lists
    return _SegmentMean.apply(feat, idx, offsets, order, m)


class _BackProjectGrad(torch.autograd.Function):
    """attaches the feature gradient to the outputs of a back-projection that already ran"""

    @staticmethod
    def forward(ctx, feats, out, mean, coords_valid, origin, voxel_size, krcam, mode):
        ctx.save_for_backward(feats)
        ctx.args = (coords_valid, origin, float(voxel_size), krcam, mode)
        ctx.has_mean = mean is not None
        return (out.view_as(out), mean.view_as(mean)) if mean is not None else (out.view_as(out), None)

    @staticmethod
    def backward(ctx, dout, dmean):
        lib = _lib.load()
        feats, = ctx.saved_tensors
        coords_valid, origin, voxel_size, krcam, mode = ctx.args
        v, b, c, h, w = feats.shape
        dev = feats.device
        nhwc = None
        if mode == BP.MODE_VARIANCE:
            nhwc = feats.detach().permute(0, 1, 3, 4, 2).contiguous()
        dout = dout.contiguous()
        dmean = dmean.contiguous() if (dmean is not None and ctx.has_mean) else None
        dfeats = torch.empty((v, b, h, w, c), dtype=torch.float32, device=dev)
        origin_f = origin.to(device=dev, dtype=torch.float32).reshape(-1, 3).contiguous()
        krcam_f = krcam.to(device=dev, dtype=torch.float32).contiguous()
        # (the deterministic form: 64-bit fixed-point accumulation with integer atomics, converted at the end)
        ws = _lib.workspace(lib.eprecon_back_project_backward_workspace_bytes(b, v, c, h, w), dev)
        _lib.check(lib.eprecon_back_project_backward_det_async(
            _lib.ptr(coords_valid), coords_valid.shape[0], _lib.ptr(origin_f), b, voxel_size, _lib.ptr(nhwc), _lib.ptr(krcam_f), v, c,
            h, w, mode, _lib.ptr(dout), dout.stride(0), _lib.ptr(dmean), _lib.ptr(dfeats), _lib.ptr(ws), ws.numel(),
            _lib.current_stream()), "eprecon_back_project_backward_det_async")
        return dfeats.permute(0, 1, 4, 2, 3), None, None, None, None, None, None, None


def back_project(coords, origin, voxel_size, feats, krcam, min_view, mode=BP.MODE_MEAN, min_valid_per_batch=1, want_mean=False):
    """back_project.run(...) whose 'feats' (and 'mean') carry the gradient with respect to `feats`"""
    with torch.no_grad():
        res = BP.run(coords, origin, voxel_size, feats, krcam, min_view, mode, min_valid_per_batch, want_mean=want_mean)
    if res is None or not _needs_grad(feats):
        return res
    out, mean = _BackProjectGrad.apply(feats, res["feats"], res.get("mean"), res["coords"].contiguous(), origin, voxel_size, krcam,
                                       mode)
    res["feats"] = out
    if mean is not None:
        res["mean"] = mean
    return res
