"""TEST INFRASTRUCTURE (see oracle/__init__.py): numpy restatement of the backward of the sparse operators, i.e. of what
csrc/backward.hip computes.  The reference obtains these gradients from torchsparse / spconv autograd
(main.py:297-313, loss.backward()); their definition is the adjoint of the forward operators restated in oracle/sparse.py
and oracle/pointvoxel.py, which is what the CPU tests check against torch autograd of a dense formulation.

  y[i] = b + sum_k x[nbr[k][i]] @ W[k]
    dx[j]  = sum_k dy[inv[k][j]] @ W[k]^T      with inv[k][j] = i  <=>  nbr[k][i] = j   (a kernel map is injective per offset)
    dW[k]  = sum_i x[nbr[k][i]]^T dy[i]
    db     = sum_i dy[i]
  out[p] = sum_c w8[p, c] * feat[idx8[p, c]]        ->  dfeat[v] = sum over (p, c) with idx8[p, c] = v of w8[p, c] * dout[p]
  mean[v] = sum_{p: idx[p] = v} feat[p] / count[v]  ->  dfeat[p] = dmean[idx[p]] / count[idx[p]]
"""
import numpy as np


def invert_map(nbr, n_in):
    """int32[K, n_out] -> int32[K, n_in]"""
    inv = np.full((nbr.shape[0], n_in), -1, np.int32)
    for k in range(nbr.shape[0]):
        live = np.flatnonzero(nbr[k] >= 0)
        inv[k, nbr[k, live]] = live
    return inv


def conv_backward(x, weight, nbr, dy, bias=False):
    """-> (dx f32[n_in, Cin], dW f32[K, Cin, Cout], db f32[Cout] | None)"""
    x, dy, w = np.asarray(x, np.float32), np.asarray(dy, np.float32), np.asarray(weight, np.float32)
    inv = invert_map(nbr, x.shape[0])
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for k in range(w.shape[0]):
        j = inv[k]
        m = j >= 0
        if m.any():
            dx[m] += dy[j[m]] @ w[k].T
        i = nbr[k]
        mo = i >= 0
        if mo.any():
            dw[k] = x[i[mo]].T @ dy[mo]
    return dx, dw, (dy.sum(0) if bias else None)


def devoxelize_backward(dout, idx8, w8, n_voxels):
    dfeat = np.zeros((n_voxels, dout.shape[1]), np.float32)
    for c in range(8):
        m = idx8[:, c] >= 0
        np.add.at(dfeat, idx8[m, c], w8[m, c, None] * dout[m])
    return dfeat


def segment_mean_backward(dmean, idx, n_voxels):
    counts = np.bincount(idx[idx >= 0], minlength=n_voxels).astype(np.float32)
    scale = np.where(counts > 0, 1.0 / np.maximum(counts, 1.0), 0.0).astype(np.float32)
    out = np.zeros((idx.shape[0], dmean.shape[1]), np.float32)
    m = idx >= 0
    out[m] = dmean[idx[m]] * scale[idx[m], None]
    return out
