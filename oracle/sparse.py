"""numpy restatement of the sparse-layer primitives (test infrastructure, see oracle/__init__.py).

PARITY UNPINNED: the reference delegates these to torchsparse (README.md:19 pins v2.0.0 but the
call sites use the 1.4-era API) and spconv (requirements.txt:18, unpinned); neither is vendored nor
installable here and the reference has no tests.  The semantics below restate the published
behaviour of those libraries as constrained by the reference's call sites
(ops/torchsparse_utils.py:15-105, models/modules.py:15-222,224-271,401-482) — SURVEY.md appendix A
— and are this build's specification.  Where the upstream order is arbitrary ("ascending hash"),
this build fixes FIRST-OCCURRENCE order.
"""
import numpy as np

BIAS = 1 << 19


def floor_div(a, q):
    return np.floor_divide(a, q)


def quantise(coords, q):
    c = np.array(coords, dtype=np.int64, copy=True)
    if q > 1:
        c[:, 1:] = floor_div(c[:, 1:], q) * q
    return c


def pack(coords):
    c = np.asarray(coords, dtype=np.int64)
    return (c[:, 0] << 60) | ((c[:, 1] + BIAS) << 40) | ((c[:, 2] + BIAS) << 20) | (c[:, 3] + BIAS)


class Index:
    """exact key -> row lookup over a coordinate set (rows = first occurrences)"""

    def __init__(self, coords, values=None):
        keys = pack(coords)
        order = np.argsort(keys, kind="stable")
        self.keys = keys[order]
        vals = np.arange(len(keys)) if values is None else np.asarray(values)
        self.vals = vals[order]
        # duplicates: keep the first (smallest row) of each run
        keep = np.ones(len(keys), bool)
        keep[1:] = self.keys[1:] != self.keys[:-1]
        self.keys, self.vals = self.keys[keep], self.vals[keep]

    def lookup(self, coords):
        k = pack(coords)
        pos = np.searchsorted(self.keys, k)
        pos = np.clip(pos, 0, max(len(self.keys) - 1, 0))
        hit = (len(self.keys) > 0) & (self.keys[pos] == k) if len(self.keys) else np.zeros(len(k), bool)
        return np.where(hit, self.vals[pos], -1).astype(np.int32)


def unique_first(coords, q=1):
    """-> (unique_coords int32[M,4] in first-occurrence order, inverse int32[N])"""
    qc = quantise(coords, q)
    keys = pack(qc)
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(first), np.int64)
    rank[order] = np.arange(len(first))
    return qc[first[order]].astype(np.int32), rank[inv].astype(np.int32)


def offsets(ksize):
    if ksize == 3:  # x fastest
        return np.array([(k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1) for k in range(27)], np.int64)
    return np.array([((k >> 2) & 1, (k >> 1) & 1, k & 1) for k in range(8)], np.int64)  # z fastest


def kernel_map(ref_coords, query_coords, ksize, stride):
    """nbr int32[K, n]: row in ref_coords of query_coords[i] + offset_k * stride, or -1"""
    idx = Index(ref_coords)
    qc = np.asarray(query_coords, np.int64)
    out = []
    for off in offsets(ksize):
        c = qc.copy()
        c[:, 1:] += off[None] * stride
        out.append(idx.lookup(c))
    return np.stack(out).astype(np.int32)


def transpose_map(fine_coords, parent, fine_stride):
    fc = np.asarray(fine_coords, np.int64)
    q = 2 * fine_stride
    b = (fc[:, 1:] - floor_div(fc[:, 1:], q) * q) // fine_stride
    kk = 4 * b[:, 0] + 2 * b[:, 1] + b[:, 2]
    up = np.full((8, len(fc)), -1, np.int32)
    up[kk, np.arange(len(fc))] = parent
    return up


def sparse_conv(x, nbr, weight, bias=None, n_out=None):
    """out[i] = bias + sum_k x[nbr[k][i]] @ W[k]; nbr None = identity (k = 1)"""
    x = np.asarray(x, np.float32)
    w = np.asarray(weight, np.float32)
    if nbr is None:
        out = x @ w.reshape(w.shape[-2], w.shape[-1])
    else:
        n_out = nbr.shape[1]
        out = np.zeros((n_out, w.shape[-1]), np.float32)
        for k in range(nbr.shape[0]):
            j = nbr[k]
            m = j >= 0
            if m.any():
                out[m] += x[j[m]] @ w[k]
    if bias is not None:
        out = out + np.asarray(bias, np.float32)[None]
    return out.astype(np.float32)


def batchnorm_train(x, gamma=None, beta=None, eps=1e-5, residual=None, relu=False):
    x64 = np.asarray(x, np.float64)
    mean = x64.mean(0)
    var = ((x64 - mean) ** 2).mean(0)
    y = (x64 - mean) / np.sqrt(var + eps)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    if residual is not None:
        y = y + residual
    if relu:
        y = np.maximum(y, 0)
    return y.astype(np.float32)


def layernorm_rows(x, gamma=None, beta=None, eps=1e-5, residual=None, pre_relu=False, post_relu=False):
    t = np.asarray(x, np.float64)
    if pre_relu:
        t = np.maximum(t, 0)
    if residual is not None:
        t = t + residual
    mean = t.mean(1, keepdims=True)
    var = ((t - mean) ** 2).mean(1, keepdims=True)
    y = (t - mean) / np.sqrt(var + eps)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    if post_relu:
        y = np.maximum(y, 0)
    return y.astype(np.float32)
