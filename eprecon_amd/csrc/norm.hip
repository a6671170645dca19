// Normalisation epilogues of the sparse layers for gfx950: batch-statistic BatchNorm over all active
// voxels (the reference runs in train mode at test time, main.py:357, so the statistics cannot be
// folded into the weights) and row-wise LayerNorm with the ReLU / residual patterns the reference
// wires around its spconv layers.
//
// Replaces  spnn.BatchNorm / nn.BatchNorm1d (train mode)   models/modules.py:22,41,54,60,65,134,139
//                                                          models/occupancy_initialization.py:29,37
//           nn.LayerNorm + ReLU + residual epilogues       models/modules.py:447-452,473-482
//                                                          models/occupancy_initialization.py:141-169
// Bandwidth-bound column / row reductions; deterministic (fixed-order partials, no float atomics).
#include "common.hpp"

namespace {
using namespace ep;

constexpr int kBnRows = 1024;  // rows per block in the column-reduction passes

// partial[blk][c] = sum over the block's rows of f(x[r][c]);  SQ: f = (x - mean[c])^2, else f = x
template <bool SQ>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *x, int n, int C, int ld,
                                                         const float *mean, float *partial)
{
    __shared__ float sRed[256];
    const int tid = threadIdx.x;
    const int rpi = 256 / C;  // rows per iteration (C <= 256)
    const int col = tid % C, rsub = tid / C;
    const bool active = rsub < rpi;
    const int r0 = blockIdx.x * kBnRows;
    const int r1 = min(n, r0 + kBnRows);
    const float mu = (SQ && active) ? mean[col] : 0.0f;
    float acc = 0.0f;
    if (active) {
        for (int r = r0 + rsub; r < r1; r += rpi) {
            const float v = x[(size_t)r * ld + col];
            if (SQ) {
                const float d = v - mu;
                acc = fmaf(d, d, acc);
            } else {
                acc += v;
            }
        }
    }
    sRed[tid] = active ? acc : 0.0f;
    __syncthreads();
    if (tid < C) {
        float s = 0.0f;
        for (int k = 0; k < rpi; ++k) s += sRed[k * C + tid];
        partial[(size_t)blockIdx.x * C + tid] = s;
    }
}

// stat[c] = (sum_blk partial[blk][c]) / n
__global__ void bn_finalize_kernel(const float *partial, int nblk, int C, int n, float *stat)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.0f;
    for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * C + c];
    stat[c] = s / (float)n;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *x, int n, int C, int ld_x,
                                                       const float *mean, const float *var,
                                                       const float *gamma, const float *beta, float eps,
                                                       const float *res, int ld_res, int relu,
                                                       float *out, int ld_out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * C) return;
    const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
    const float inv = 1.0f / sqrtf(var[c] + eps);
    float v = (x[(size_t)r * ld_x + c] - mean[c]) * inv;
    v = v * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f);
    if (res) v += res[(size_t)r * ld_res + c];
    if (relu) v = fmaxf(v, 0.0f);
    out[(size_t)r * ld_out + c] = v;
}

// LayerNorm over the C channels of each row: t = x; [relu]; [+ res]; LN(t) * g + b; [relu]
// 8 lanes per row, lane g owns channels g, g+8, ...
__global__ __launch_bounds__(256) void rowwise_ln_kernel(const float *x, int n, int C, int ld_x,
                                                         const float *res, int ld_res,
                                                         const float *gamma, const float *beta, float eps,
                                                         int pre_relu, int post_relu, float *out,
                                                         int ld_out)
{
    const int g = threadIdx.x & 7;
    const int r = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = r < n;
    const float *xr = x + (size_t)(live ? r : 0) * ld_x;
    const float *rr = res ? res + (size_t)(live ? r : 0) * ld_res : nullptr;
    auto value = [&](int c) -> float {
        float v = xr[c];
        if (pre_relu) v = fmaxf(v, 0.0f);
        if (rr) v += rr[c];
        return v;
    };
    float s = 0.0f;
    if (live)
        for (int c = g; c < C; c += 8) s += value(c);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    const float mean = s / (float)C;
    float q = 0.0f;
    if (live)
        for (int c = g; c < C; c += 8) {
            const float d = value(c) - mean;
            q = fmaf(d, d, q);
        }
    q += __shfl_xor(q, 1);
    q += __shfl_xor(q, 2);
    q += __shfl_xor(q, 4);
    const float inv = 1.0f / sqrtf(q / (float)C + eps);
    if (live)
        for (int c = g; c < C; c += 8) {
            float v = (value(c) - mean) * inv * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f);
            if (post_relu) v = fmaxf(v, 0.0f);
            out[(size_t)r * ld_out + c] = v;
        }
}

}  // namespace

extern "C" {

size_t eprecon_batchnorm_workspace_bytes(int64_t n, int channels)
{
    const size_t nblk = (size_t)ceil_div(n > 0 ? n : 1, kBnRows);
    return align_up(nblk * channels * sizeof(float), 256) + 2 * align_up((size_t)channels * sizeof(float), 256);
}

// Train-mode BatchNorm over the n rows: mean, biased variance (two-pass), affine, optional residual
// add and ReLU.  out may alias x.  mean_out / var_out (optional, device) receive the statistics.
int eprecon_batchnorm_train_async(const float *x, int64_t n, int channels, int ld_x, const float *gamma,
                                  const float *beta, float eps, const float *residual, int ld_res,
                                  int relu, float *out, int ld_out, float *mean_out, float *var_out,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !out || n < 0 || channels <= 0 || channels > 256 || ld_x < channels || ld_out < channels ||
        !workspace)
        return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_batchnorm_workspace_bytes(n, channels)) return EPRECON_ERR_WORKSPACE;
    if (n == 0) return EPRECON_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)ceil_div(n, kBnRows);
    char *ws = reinterpret_cast<char *>(workspace);
    float *partial = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)nblk * channels * sizeof(float), 256);
    float *mean = mean_out ? mean_out : reinterpret_cast<float *>(ws);
    ws += align_up((size_t)channels * sizeof(float), 256);
    float *var = var_out ? var_out : reinterpret_cast<float *>(ws);
    const dim3 fgrid((channels + 63) / 64), fblock(64);
    hipLaunchKernelGGL((bn_partial_kernel<false>), dim3(nblk), dim3(256), 0, st, x, (int)n, channels, ld_x,
                       (const float *)nullptr, partial);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, fgrid, fblock, 0, st, partial, nblk, channels, (int)n, mean);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL((bn_partial_kernel<true>), dim3(nblk), dim3(256), 0, st, x, (int)n, channels, ld_x,
                       (const float *)mean, partial);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, fgrid, fblock, 0, st, partial, nblk, channels, (int)n, var);
    EP_LAUNCH_CHECK();
    const size_t total = (size_t)n * channels;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ceil_div((int64_t)total, 256)), dim3(256), 0, st, x,
                       (int)n, channels, ld_x, (const float *)mean, (const float *)var, gamma, beta, eps,
                       residual, ld_res, relu, out, ld_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_rowwise_layernorm_async(const float *x, int64_t n, int channels, int ld_x,
                                    const float *residual, int ld_res, const float *gamma,
                                    const float *beta, float eps, int pre_relu, int post_relu,
                                    float *out, int ld_out, void *stream)
{
    if (!x || !out || n < 0 || channels <= 0 || ld_x < channels || ld_out < channels) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(rowwise_ln_kernel, dim3((unsigned)ceil_div(n, 32)), dim3(256), 0,
                       (hipStream_t)stream, x, (int)n, channels, ld_x, residual, ld_res, gamma, beta, eps,
                       pre_relu, post_relu, out, ld_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"
