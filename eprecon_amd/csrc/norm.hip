// Normalisation epilogues of the sparse layers for gfx950: batch-statistic BatchNorm over all active
// voxels (the reference runs in train mode at test time, main.py:357, so the statistics cannot be
// folded into the weights) and row-wise LayerNorm with the ReLU / residual patterns the reference
// wires around its spconv layers.
//
// Replaces  spnn.BatchNorm / nn.BatchNorm1d (train mode)   models/modules.py:22,41,54,60,65,134,139
//                                                          models/occupancy_initialization.py:29,37
//           nn.LayerNorm + ReLU + residual epilogues       models/modules.py:447-452,473-482
//                                                          models/occupancy_initialization.py:141-169
// Bandwidth-bound column / row reductions; deterministic (fixed-order Chan merges, no float atomics).
#include <stdlib.h>

#include "common.hpp"
#include "conv_common.hpp"

namespace {
using namespace ep;

constexpr int kBnPerThread = 8;  // rows each thread of the statistics kernel keeps in registers

// Chan et al. merge of two (count, mean, M2) summaries — exact up to rounding, order fixed by the caller
__device__ __forceinline__ void chan_merge(float &n_a, float &mean_a, float &m2_a, float n_b, float mean_b, float m2_b)
{
    if (n_b == 0.0f) return;
    if (n_a == 0.0f) {
        n_a = n_b; mean_a = mean_b; m2_a = m2_b;
        return;
    }
    const float n = n_a + n_b;
    const float d = mean_b - mean_a;
    mean_a = mean_a + d * (n_b / n);
    m2_a = m2_a + m2_b + d * d * (n_a * n_b / n);
    n_a = n;
}

// One pass over x: block b summarises rows [b*R, (b+1)*R) with R = kBnPerThread * (256 / C); every
// thread owns one column and <= 8 rows held in registers (two-pass mean / M2 on those, no
// cancellation), then the (256/C) row groups are merged through LDS in fixed order.
// partial[b][0][c] = count, [1][c] = mean, [2][c] = M2.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *x, int n, int C, int ld, float *partial)
{
    __shared__ float sN[256], sMean[256], sM2[256];
    const int tid = threadIdx.x;
    const int rpi = 256 / C;
    const int col = tid % C, rsub = tid / C;
    const bool active = rsub < rpi;
    const int r0 = blockIdx.x * (kBnPerThread * rpi);
    float v[kBnPerThread];
    float cnt = 0.0f, sum = 0.0f;
#pragma unroll
    for (int k = 0; k < kBnPerThread; ++k) {
        const int r = r0 + k * rpi + rsub;
        const bool ok = active && r < n;
        v[k] = ok ? x[(size_t)r * ld + col] : 0.0f;
        cnt += ok ? 1.0f : 0.0f;
        sum += v[k];
    }
    const float mean = cnt > 0.0f ? sum / cnt : 0.0f;
    float m2 = 0.0f;
#pragma unroll
    for (int k = 0; k < kBnPerThread; ++k) {
        const int r = r0 + k * rpi + rsub;
        if (active && r < n) {
            const float d = v[k] - mean;
            m2 = fmaf(d, d, m2);
        }
    }
    sN[tid] = cnt; sMean[tid] = mean; sM2[tid] = m2;
    __syncthreads();
    // fixed-order pairwise tree over the rpi row groups (a serial merge by one thread per column cost
    // 44 us on the single-channel BatchNorm of the occupancy logit: 256 dependent merges with divisions)
    for (int s = 1; s < rpi; s <<= 1) {
        if (active && (rsub % (2 * s)) == 0 && rsub + s < rpi) {
            float n = sN[tid], m = sMean[tid], q = sM2[tid];
            const int o = tid + s * C;
            chan_merge(n, m, q, sN[o], sMean[o], sM2[o]);
            sN[tid] = n; sMean[tid] = m; sM2[tid] = q;
        }
        __syncthreads();
    }
    if (tid < C) {
        float *p = partial + (size_t)blockIdx.x * 3 * C;
        p[tid] = sN[tid]; p[C + tid] = sMean[tid]; p[2 * C + tid] = sM2[tid];
    }
}

// one workgroup per channel: thread t merges partials t, t+256, ... in order, then a fixed LDS tree
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *partial, int nblk, int C, float *mean_out,
                                                          float *var_out)
{
    __shared__ float sN[256], sMean[256], sM2[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    float a_n = 0.0f, a_mean = 0.0f, a_m2 = 0.0f;
    // (the summaries of four rows are loaded before the first is merged: the chain of load latencies, not the arithmetic, is
    // what this kernel costs; the merge order — rows t, t + 256, ... — is unchanged)
    for (int b = tid; b < nblk; b += 4 * 256) {
        float vn[4], vm[4], vq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int bb = b + u * 256;
            const float *p = partial + (size_t)(bb < nblk ? bb : b) * 3 * C;
            vn[u] = bb < nblk ? p[c] : 0.0f;
            vm[u] = p[C + c];
            vq[u] = p[2 * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) chan_merge(a_n, a_mean, a_m2, vn[u], vm[u], vq[u]);
    }
    sN[tid] = a_n; sMean[tid] = a_mean; sM2[tid] = a_m2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            float n = sN[tid], m = sMean[tid], q = sM2[tid];
            chan_merge(n, m, q, sN[tid + s], sMean[tid + s], sM2[tid + s]);
            sN[tid] = n; sMean[tid] = m; sM2[tid] = q;
        }
        __syncthreads();
    }
    if (tid == 0) {
        mean_out[c] = sMean[0];
        var_out[c] = sN[0] > 0.0f ? sM2[0] / sN[0] : 0.0f;  // biased variance
    }
}

// the same merge, published in affine form: y = x * scale + shift  ==  (x - mean) / sqrt(var + eps) * gamma + beta
__global__ __launch_bounds__(256) void bn_finalize_affine_kernel(const float *partial, int nblk, int C, const float *gamma,
                                                                 const float *beta, float eps, float *scale_out,
                                                                 float *shift_out)
{
    // one workgroup per channel; fixed merge order: thread t takes rows t, t + 256, ...; lanes merge by
    // xor-shuffles (lower lane first), the four wave results in wave order -- one barrier in total
    __shared__ float sN[4], sMean[4], sM2[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    // (gamma / beta travel with the first summaries instead of behind the merge: one memory round trip less on a launch that
    // is nothing but round trips — 37 of these sit between the layers of a cfg2 step)
    const float g_c = gamma ? gamma[c] : 1.0f, b_c = beta ? beta[c] : 0.0f;
    float a_n = 0.0f, a_mean = 0.0f, a_m2 = 0.0f;
    // (the summaries of four rows are loaded before the first is merged: the chain of load latencies, not the arithmetic, is
    // what this kernel costs; the merge order — rows t, t + 256, ... — is unchanged)
    for (int b = tid; b < nblk; b += 4 * 256) {
        float vn[4], vm[4], vq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int bb = b + u * 256;
            const float *p = partial + (size_t)(bb < nblk ? bb : b) * 3 * C;
            vn[u] = bb < nblk ? p[c] : 0.0f;
            vm[u] = p[C + c];
            vq[u] = p[2 * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) chan_merge(a_n, a_mean, a_m2, vn[u], vm[u], vq[u]);
    }
    const int lane = tid & 63;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float on = __shfl_xor(a_n, m), om = __shfl_xor(a_mean, m), oq = __shfl_xor(a_m2, m);
        // both partners compute the same (lower, upper) merge, so every lane ends with the same value
        float ln = (lane & m) ? on : a_n, lm = (lane & m) ? om : a_mean, lq = (lane & m) ? oq : a_m2;
        chan_merge(ln, lm, lq, (lane & m) ? a_n : on, (lane & m) ? a_mean : om, (lane & m) ? a_m2 : oq);
        a_n = ln; a_mean = lm; a_m2 = lq;
    }
    if (lane == 0) { sN[tid >> 6] = a_n; sMean[tid >> 6] = a_mean; sM2[tid >> 6] = a_m2; }
    __syncthreads();
    if (tid == 0) {
        float n = sN[0], mean = sMean[0], m2 = sM2[0];
        for (int w = 1; w < 4; ++w) chan_merge(n, mean, m2, sN[w], sMean[w], sM2[w]);
        const float var = n > 0.0f ? m2 / n : 0.0f;  // biased variance
        const float sc = g_c / sqrtf(var + eps);
        scale_out[c] = sc;
        shift_out[c] = b_c - mean * sc;
    }
}

// res_scale / res_shift (optional): the residual operand carries a pending BatchNorm of its own in affine form (the 1x1 skip
// convolution of a residual block, models/modules.py:57-65), applied on load
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *x, int n, int C, int ld_x,
                                                       const float *mean, const float *var,
                                                       const float *gamma, const float *beta, float eps,
                                                       const float *res, int ld_res, const float *res_scale,
                                                       const float *res_shift, int relu, float *out, int ld_out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * C) return;
    const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
    const float inv = 1.0f / sqrtf(var[c] + eps);
    float v = (x[(size_t)r * ld_x + c] - mean[c]) * inv;
    v = v * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f);
    if (res) {
        float rv = res[(size_t)r * ld_res + c];
        if (res_scale) rv = fmaf(rv, res_scale[c], res_shift[c]);
        v += rv;
    }
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

__global__ __launch_bounds__(256) void affine_rows_kernel(const float *x, int n, int C, int ld_x, const float *scale,
                                                          const float *shift, int relu, float *out, int ld_out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * C) return;
    const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
    float v = fmaf(x[(size_t)r * ld_x + c], scale[c], shift[c]);
    if (relu) v = fmaxf(v, 0.0f);
    out[(size_t)r * ld_out + c] = v;
}

// BatchNorm form (c) (csrc/conv_common.hpp): the accumulator block of a producer -> (scale, shift) vectors — the stand-alone
// finish for consumers that do not do it in their prologue (dense-grid tile kernels, residual operands)
__global__ __launch_bounds__(256) void bn_acc_affine_kernel(const long long *acc, int ld, int c0, int C, float eps, float *scale,
                                                            float *shift)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float sc, sh;
    epconv::bn_acc_affine(acc, ld, c0 + c, eps, sc, sh);
    scale[c] = sc;
    shift[c] = sh;
}

// ... and the materialising consumer: out = [relu](x * scale + shift), every workgroup finishing the block itself (C <= 512)
constexpr int kAccRowsMaxC = 512;
__global__ __launch_bounds__(256) void affine_rows_acc_kernel(const float *x, int n, int C, int ld_x, const long long *acc, int ld,
                                                              int c0, float eps, int relu, float *out, int ld_out)
{
    __shared__ float sAff[2 * kAccRowsMaxC];
    for (int c = threadIdx.x; c < C; c += 256) epconv::bn_acc_affine(acc, ld, c0 + c, eps, sAff[c], sAff[kAccRowsMaxC + c]);
    __syncthreads();
    // a workgroup applies them to 16 x 256 elements
    for (int k = 0; k < 16; ++k) {
        const size_t e = ((size_t)blockIdx.x * 16 + k) * 256 + threadIdx.x;
        if (e >= (size_t)n * C) return;
        const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
        float v = fmaf(x[(size_t)r * ld_x + c], sAff[c], sAff[kAccRowsMaxC + c]);
        if (relu) v = fmaxf(v, 0.0f);
        out[(size_t)r * ld_out + c] = v;
    }
}

__global__ __launch_bounds__(256) void affine_rows_res_kernel(const float *x, int n, int C, int ld_x, const float *scale,
                                                              const float *shift, const float *res, int ld_res, int relu,
                                                              float *out, int ld_out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * C) return;
    const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
    float v = fmaf(x[(size_t)r * ld_x + c], scale[c], shift[c]);
    if (res) v += res[(size_t)r * ld_res + c];
    if (relu) v = fmaxf(v, 0.0f);
    out[(size_t)r * ld_out + c] = v;
}

// (One-launch forms for short summary lists — every workgroup finishing the statistics itself before applying them — were
// measured twice and removed: round 3 with one wave per channel (a 10 us serial chain of Chan merges per workgroup), round 4
// with 256 / C row groups merging side by side and the launch limited to ~16 MB of redundant summary reads: SPVCNN still lost
// 0.1-0.18 ms per level against the separate 5 us finalize launch, gpurun r04_c / r04_d.  A third form — workgroups of
// 64 channels x 256 rows, the list merged once per workgroup by four row lanes with eight summary rows in flight — lost as well:
// SPVCNN of the coarsest level 1.42 -> 1.68 ms.  The merge is a chain of dependent divisions; 74 workgroups repeating it in
// front of their rows is slower than one 5 us launch doing it once.)
int bn_finalize_apply(const float *x, int64_t n, int channels, int ld_x, const float *partial, int nblk,
                      const float *gamma, const float *beta, float eps, const float *residual, int ld_res,
                      int relu, float *out, int ld_out, float *mean, float *var, hipStream_t st,
                      const float *res_scale = nullptr, const float *res_shift = nullptr)
{
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(channels), dim3(256), 0, st, partial, nblk, channels, mean, var);
    EP_LAUNCH_CHECK();
    const size_t total = (size_t)n * channels;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ceil_div((int64_t)total, 256)), dim3(256), 0, st, x,
                       (int)n, channels, ld_x, (const float *)mean, (const float *)var, gamma, beta, eps,
                       residual, ld_res, res_scale, res_shift, relu, out, ld_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // namespace

extern "C" {

static inline int bn_rows_per_block(int channels) { return kBnPerThread * (256 / channels); }

size_t eprecon_batchnorm_workspace_bytes(int64_t n, int channels)
{
    if (channels <= 0 || channels > 256) return 0;
    const size_t nblk = (size_t)ceil_div(n > 0 ? n : 1, bn_rows_per_block(channels));
    return align_up(nblk * 3 * channels * sizeof(float), 256) + 2 * align_up((size_t)channels * sizeof(float), 256);
}

// Train-mode BatchNorm over the n rows: one statistics pass (per-block (count, mean, M2) merged
// with Chan's formula in fixed order -> deterministic, no cancellation), affine, optional residual
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
    const int nblk = (int)ceil_div(n, bn_rows_per_block(channels));
    char *ws = reinterpret_cast<char *>(workspace);
    float *partial = reinterpret_cast<float *>(ws);
    ws += align_up((size_t)nblk * 3 * channels * sizeof(float), 256);
    float *mean = mean_out ? mean_out : reinterpret_cast<float *>(ws);
    ws += align_up((size_t)channels * sizeof(float), 256);
    float *var = var_out ? var_out : reinterpret_cast<float *>(ws);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(256), 0, st, x, (int)n, channels, ld_x, partial);
    EP_LAUNCH_CHECK();
    return bn_finalize_apply(x, n, channels, ld_x, partial, nblk, gamma, beta, eps, residual, ld_res, relu, out,
                             ld_out, mean, var, st);
}

size_t eprecon_batchnorm_apply_workspace_bytes(int channels)
{
    return 2 * align_up((size_t)(channels > 0 ? channels : 1) * sizeof(float), 256);
}

// Second half of the train-mode BatchNorm for a tensor whose per-block (count, mean, M2) summaries
// were already produced by its producer (eprecon_sparse_conv_fused_async's bn_partial,
// [nblk][3][channels]): fixed-order merge -> mean / biased variance -> affine [+ residual] [ReLU].
int eprecon_batchnorm_apply_partials_async(const float *x, int64_t n, int channels, int ld_x,
                                           const float *partial, int64_t nblk, const float *gamma,
                                           const float *beta, float eps, const float *residual,
                                           int ld_res, int relu, float *out, int ld_out, float *mean_out,
                                           float *var_out, void *workspace, size_t workspace_bytes,
                                           void *stream)
{
    if (!x || !out || !partial || n < 0 || nblk <= 0 || nblk > 0x7fffffff || channels <= 0 || ld_x < channels ||
        ld_out < channels || !workspace)
        return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_batchnorm_apply_workspace_bytes(channels)) return EPRECON_ERR_WORKSPACE;
    if (n == 0) return EPRECON_OK;
    char *ws = reinterpret_cast<char *>(workspace);
    float *mean = mean_out ? mean_out : reinterpret_cast<float *>(ws);
    ws += align_up((size_t)channels * sizeof(float), 256);
    float *var = var_out ? var_out : reinterpret_cast<float *>(ws);
    return bn_finalize_apply(x, n, channels, ld_x, partial, (int)nblk, gamma, beta, eps, residual, ld_res, relu,
                             out, ld_out, mean, var, (hipStream_t)stream);
}

int eprecon_batchnorm_apply_partials_res_async(const float *x, int64_t n, int channels, int ld_x, const float *partial,
                                               int64_t nblk, const float *gamma, const float *beta, float eps,
                                               const float *residual, int ld_res, const float *res_scale,
                                               const float *res_shift, int relu, float *out, int ld_out, void *workspace,
                                               size_t workspace_bytes, void *stream)
{
    if (!x || !out || !partial || !residual || !res_scale || !res_shift || n < 0 || nblk <= 0 || nblk > 0x7fffffff ||
        channels <= 0 || ld_x < channels || ld_out < channels || ld_res < channels || !workspace)
        return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_batchnorm_apply_workspace_bytes(channels)) return EPRECON_ERR_WORKSPACE;
    if (n == 0) return EPRECON_OK;
    char *ws = reinterpret_cast<char *>(workspace);
    float *mean = reinterpret_cast<float *>(ws);
    float *var = reinterpret_cast<float *>(ws + align_up((size_t)channels * sizeof(float), 256));
    return bn_finalize_apply(x, n, channels, ld_x, partial, (int)nblk, gamma, beta, eps, residual, ld_res, relu, out, ld_out,
                             mean, var, (hipStream_t)stream, res_scale, res_shift);
}

int eprecon_batchnorm_finalize_affine_async(const float *partial, int64_t nblk, int channels, const float *gamma,
                                            const float *beta, float eps, float *scale_out, float *shift_out,
                                            void *stream)
{
    if (!partial || !scale_out || !shift_out || nblk <= 0 || nblk > 0x7fffffff || channels <= 0) return EPRECON_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_affine_kernel, dim3(channels), dim3(256), 0, (hipStream_t)stream, partial, (int)nblk,
                       channels, gamma, beta, eps, scale_out, shift_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_affine_rows_async(const float *x, int64_t n, int channels, int ld_x, const float *scale,
                              const float *shift, int relu, float *out, int ld_out, void *stream)
{
    if (!x || !out || !scale || !shift || n < 0 || channels <= 0 || ld_x < channels || ld_out < channels ||
        n * channels > 0x7fffffffll * 256)
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(affine_rows_kernel, dim3((unsigned)ceil_div(n * channels, (int64_t)256)), dim3(256), 0,
                       (hipStream_t)stream, x, (int)n, channels, ld_x, scale, shift, relu, out, ld_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_batchnorm_acc_affine_async(const long long *acc, int acc_ld, int acc_c0, int channels, float eps, float *scale_out,
                                       float *shift_out, void *stream)
{
    if (!acc || !scale_out || !shift_out || channels <= 0 || acc_c0 < 0 || acc_c0 + channels > acc_ld) return EPRECON_ERR_ARG;
    hipLaunchKernelGGL(bn_acc_affine_kernel, dim3((unsigned)ceil_div(channels, 256)), dim3(256), 0, (hipStream_t)stream, acc, acc_ld,
                       acc_c0, channels, eps, scale_out, shift_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_affine_rows_acc_async(const float *x, int64_t n, int channels, int ld_x, const long long *acc, int acc_ld, int acc_c0,
                                  float eps, int relu, float *out, int ld_out, void *stream)
{
    if (!x || !out || !acc || n < 0 || channels <= 0 || channels > kAccRowsMaxC || ld_x < channels || ld_out < channels ||
        acc_c0 < 0 || acc_c0 + channels > acc_ld || n * channels > 0x7fffffffll * 256)
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(affine_rows_acc_kernel, dim3((unsigned)ceil_div(n * channels, (int64_t)4096)), dim3(256), 0,
                       (hipStream_t)stream, x, (int)n, channels, ld_x, acc, acc_ld, acc_c0, eps, relu, out, ld_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_affine_rows_res_async(const float *x, int64_t n, int channels, int ld_x, const float *scale,
                                  const float *shift, const float *residual, int ld_res, int relu, float *out,
                                  int ld_out, void *stream)
{
    if (!x || !out || !scale || !shift || n < 0 || channels <= 0 || ld_x < channels || ld_out < channels ||
        (residual && ld_res < channels) || n * channels > 0x7fffffffll * 256)
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(affine_rows_res_kernel, dim3((unsigned)ceil_div(n * channels, (int64_t)256)), dim3(256), 0,
                       (hipStream_t)stream, x, (int)n, channels, ld_x, scale, shift, residual, ld_res, relu, out, ld_out);
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
