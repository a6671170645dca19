// The per-voxel prediction heads as ONE launch (gfx950): Linear(C, 4C) - LayerNorm - ReLU - Linear(4C, C) - LayerNorm - ReLU -
// Linear(C, C_out) (+ skip when C == C_out) — the reference's Linear4xTrans (models/modules.py:273-311), which NeuConNet runs
// per level for the TSDF head, the occupancy head (models/neucon_network.py:437-438: both on the same rows) and the panoptic
// projection (:546-548).  As separate launches a head is 3 GEMMs + 2 row-wise LayerNorms and writes / re-reads the 4C-wide
// intermediate of every voxel (371k x 96 floats at the finest level); here a wave owns 16 voxels and the whole chain stays in
// its registers:
//   the products are taken TRANSPOSED, features x voxels: v_mfma_f32_16x16x4_f32 with A = a 16-feature x 4-input tile of the
//   weights and B = 4 inputs x 16 voxels.  Its result registers hold, in lane (voxel l16, group q), features 16 t + 4 q + i of
//   that voxel (tile t, register i) — which is exactly the B operand of the NEXT product when its A tiles are packed with the
//   same input order (input 16 c + 4 q + i for MFMA i of chunk c): no transpose, no LDS, no barrier between the layers.
//   A LayerNorm is a sum over a lane's registers and over its 4 q-groups (two cross-lane adds).
// Weights come packed per (output tile, input chunk) as 1 KB blocks (lane -> float4 over i), read through L1 / L2 by every wave
// in the same order.  blockIdx.y selects the head (TSDF and occupancy share the launch and the L2-resident rows).
// Sums: k ascending inside an MFMA chain, the LayerNorm in two passes (mean, then squared deviations): equal to the PyTorch
// modules within fp32 round-off; deterministic.
#include "common.hpp"

namespace {
using namespace ep;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MlpParams {
    const float *x;
    int ldx, n, C, Cout, residual, vec4;
    float eps1, eps2;
    const float *w1[2], *b1[2], *g1[2], *be1[2];   // packed [T1][KC][64][4]; vectors padded to 16 T1
    const float *w2[2], *b2[2], *g2[2], *be2[2];   // packed [T2][T1][64][4]; vectors padded to 16 T2 (zeros behind C)
    const float *w3[2], *b3[2];                    // packed [T3][T2][64][4]; bias padded to 16 T3
    float *y[2];
    int ldy[2];
};

__device__ __forceinline__ f32x4 ld4(const float *p)
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    return f32x4{v.x, v.y, v.z, v.w};
}

// y = [relu]( (a - mean) / sqrt(var + eps) * g + b ) over the `count` real features of a voxel: the lane's T tiles x 4
// registers and the 4 lane groups q.  Features >= count (padding of the last tile) are left out of the sums and come out 0
// (their g / b are stored as 0).
template <int T>
__device__ __forceinline__ void layernorm_relu(f32x4 (&a)[T], int q, int count, float eps, const float *g, const float *b)
{
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += (16 * t + 4 * q + i < count) ? a[t][i] : 0.0f;
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)count;
    float v = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = (16 * t + 4 * q + i < count) ? a[t][i] - mean : 0.0f;
            v = fmaf(d, d, v);
        }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const float inv = 1.0f / sqrtf(v / (float)count + eps);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const f32x4 gg = ld4(g + 16 * t + 4 * q), bb = ld4(b + 16 * t + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[t][i] = fmaxf((a[t][i] - mean) * inv * gg[i] + bb[i], 0.0f);
    }
}

// out[t] = bias tile t + sum over the KIN input chunks: A = packed weights (block (t, c)), B = in[c]
template <int TOUT, int KIN>
__device__ __forceinline__ void layer(f32x4 (&out)[TOUT], const f32x4 (&in)[KIN], const float *wp, const float *bias, int lane, int q)
{
#pragma unroll
    for (int t = 0; t < TOUT; ++t) out[t] = ld4(bias + 16 * t + 4 * q);
    const float *w = wp + lane * 4;
#pragma unroll
    for (int c = 0; c < KIN; ++c) {
        f32x4 a[TOUT];
#pragma unroll
        for (int t = 0; t < TOUT; ++t) a[t] = ld4(w + (size_t)(t * KIN + c) * 256);
        // (the TOUT accumulators of a step are independent: back-to-back MFMAs never wait on each other)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < TOUT; ++t) out[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][i], in[c][i], out[t], 0, 0, 0);
    }
}

template <int KC, int T1, int T3>
__global__ __launch_bounds__(256) void mlp4x_kernel(MlpParams p)
{
    constexpr int T2 = KC;
    const int head = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, q = lane >> 4;
    const int v0 = ((int)blockIdx.x * 4 + wave) * 16;
    if (v0 >= p.n) return;                       // (no barrier anywhere in the kernel)
    const int vox = min(v0 + l16, p.n - 1);
    const float *xr = p.x + (size_t)vox * p.ldx;
    f32x4 xb[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int ch = 16 * c + 4 * q;
        xb[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (ch < p.C) {                           // (C % 4 == 0)
            if (p.vec4) xb[c] = ld4(xr + ch);
            else xb[c] = f32x4{xr[ch], xr[ch + 1], xr[ch + 2], xr[ch + 3]};
        }
    }
    f32x4 h1[T1];
    layer<T1, KC>(h1, xb, p.w1[head], p.b1[head], lane, q);
    layernorm_relu<T1>(h1, q, 4 * p.C, p.eps1, p.g1[head], p.be1[head]);
    f32x4 h2[T2];
    layer<T2, T1>(h2, h1, p.w2[head], p.b2[head], lane, q);
    layernorm_relu<T2>(h2, q, p.C, p.eps2, p.g2[head], p.be2[head]);
    f32x4 o[T3];
    layer<T3, T2>(o, h2, p.w3[head], p.b3[head], lane, q);
    if (v0 + l16 >= p.n) return;
    float *yr = p.y[head] + (size_t)(v0 + l16) * p.ldy[head];
#pragma unroll
    for (int t = 0; t < T3; ++t) {
        const int f = 16 * t + 4 * q;
        if constexpr (T3 == T2) {
            if (p.residual) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[t][i] += h2[t][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (f + i < p.Cout) yr[f + i] = o[t][i];
    }
}

// Short lists (a few thousand voxels: 147 workgroups of the form above at the coarsest level, every wave streaming the whole
// 300 KB of weights behind a chain of L2 latencies): the FOUR waves of a workgroup share 16 voxels.  Wave w takes a quarter of
// the hidden (4C) features through the first product and LayerNorm (its statistics exchanged through LDS), multiplies its
// quarter of the second product's inputs — a K-split, the four partial results summed in wave order through LDS — and the
// tail (second LayerNorm, last product) is replicated, wave t storing output tile t.  4 x the workgroups, a quarter of the
// weight stream per wave.
template <int KC, int T1, int T3>
__global__ __launch_bounds__(256) void mlp4x_split_kernel(MlpParams p)
{
    constexpr int T2 = KC, TW = (T1 + 3) / 4;
    __shared__ float sRed[2][4][16];
    __shared__ float sPart[4][T2][256];
    const int head = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, q = lane >> 4;
    const int v0 = (int)blockIdx.x * 16;
    const int vox = min(v0 + l16, p.n - 1);
    const float *xr = p.x + (size_t)vox * p.ldx;
    f32x4 xb[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int ch = 16 * c + 4 * q;
        xb[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (ch < p.C) {
            if (p.vec4) xb[c] = ld4(xr + ch);
            else xb[c] = f32x4{xr[ch], xr[ch + 1], xr[ch + 2], xr[ch + 3]};
        }
    }
    const int t0 = wave * T1 / 4, t1 = (wave + 1) * T1 / 4;     // this wave's hidden tiles (uniform)
    // ---- first product on the wave's tiles ----
    f32x4 h1[TW];
    {
        const float *w = p.w1[head] + lane * 4;
#pragma unroll
        for (int j = 0; j < TW; ++j) h1[j] = t0 + j < t1 ? ld4(p.b1[head] + 16 * (t0 + j) + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            f32x4 a[TW];
#pragma unroll
            for (int j = 0; j < TW; ++j) a[j] = ld4(w + (size_t)((t0 + min(j, t1 - t0 - 1)) * KC + c) * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
                    if (t0 + j < t1) h1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], xb[c][i], h1[j], 0, 0, 0);
        }
    }
    // ---- LayerNorm over all 4C features: partial sums per wave, totals through LDS (wave order) ----
    const float cnt = (float)(4 * p.C);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < TW; ++j)
        if (t0 + j < t1) s += h1[j][0] + h1[j][1] + h1[j][2] + h1[j][3];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (q == 0) sRed[0][wave][l16] = s;
    __syncthreads();
    const float mean = (sRed[0][0][l16] + sRed[0][1][l16] + sRed[0][2][l16] + sRed[0][3][l16]) / cnt;
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < TW; ++j)
        if (t0 + j < t1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = h1[j][i] - mean;
                v = fmaf(d, d, v);
            }
        }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (q == 0) sRed[1][wave][l16] = v;
    __syncthreads();
    const float inv = 1.0f / sqrtf((sRed[1][0][l16] + sRed[1][1][l16] + sRed[1][2][l16] + sRed[1][3][l16]) / cnt + p.eps1);
#pragma unroll
    for (int j = 0; j < TW; ++j)
        if (t0 + j < t1) {
            const f32x4 gg = ld4(p.g1[head] + 16 * (t0 + j) + 4 * q), bb = ld4(p.be1[head] + 16 * (t0 + j) + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) h1[j][i] = fmaxf((h1[j][i] - mean) * inv * gg[i] + bb[i], 0.0f);
        }
    // ---- second product: this wave's share of the inputs, partial results to LDS ----
    {
        f32x4 part[T2];
#pragma unroll
        for (int t = 0; t < T2; ++t) part[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *w = p.w2[head] + lane * 4;
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            if (t0 + j < t1) {
                f32x4 a[T2];
#pragma unroll
                for (int t = 0; t < T2; ++t) a[t] = ld4(w + (size_t)(t * T1 + t0 + j) * 256);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < T2; ++t) part[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][i], h1[j][i], part[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < T2; ++t) *reinterpret_cast<f32x4 *>(&sPart[wave][t][lane * 4]) = part[t];
    }
    __syncthreads();
    f32x4 h2[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t) {
        h2[t] = ld4(p.b2[head] + 16 * t + 4 * q);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4 o = *reinterpret_cast<const f32x4 *>(&sPart[w][t][lane * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i) h2[t][i] += o[i];
        }
    }
    layernorm_relu<T2>(h2, q, p.C, p.eps2, p.g2[head], p.be2[head]);
    // ---- last product: output tile `wave` ----
    if (wave >= T3) return;                       // (wave-uniform: the MFMAs below need every lane of the wave)
    f32x4 o = ld4(p.b3[head] + 16 * wave + 4 * q);
    {
        const float *w = p.w3[head] + lane * 4;
#pragma unroll
        for (int c = 0; c < T2; ++c) {
            const f32x4 a = ld4(w + (size_t)(wave * T2 + c) * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], h2[c][i], o, 0, 0, 0);
        }
    }
    if (v0 + l16 >= p.n) return;
    float *yr = p.y[head] + (size_t)(v0 + l16) * p.ldy[head];
    const int f = 16 * wave + 4 * q;
    if constexpr (T3 == T2) {
        if (p.residual) {
            // (h2 tile `wave` of this lane: a compile-time index is needed for the register array)
#pragma unroll
            for (int t = 0; t < T2; ++t)
                if (t == wave) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] += h2[t][i];
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (f + i < p.Cout) yr[f + i] = o[i];
}

template <int KC, int T1, int T3>
int launch(const MlpParams &p, int heads, hipStream_t st)
{
    // the four-waves-per-16-voxels form while the one-wave form would leave most CUs without a second workgroup
    if (p.n <= 40000)
        hipLaunchKernelGGL((mlp4x_split_kernel<KC, T1, T3>), dim3((unsigned)ceil_div(p.n, (int64_t)16), (unsigned)heads), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((mlp4x_kernel<KC, T1, T3>), dim3((unsigned)ceil_div(p.n, (int64_t)64), (unsigned)heads), dim3(256), 0, st, p);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // namespace

extern "C" {

int eprecon_mlp4x_supported(int channels, int out_channels)
{
    const bool head = out_channels >= 1 && out_channels <= 16 && (channels == 24 || channels == 48 || channels == 96);
    const bool proj = out_channels > 32 && out_channels <= 48 && (channels == 48 || channels == 88 || channels == 176);
    return head || proj;
}

int eprecon_mlp4x_async(const eprecon_mlp4x_desc *d, void *stream)
{
    if (!d || d->n < 0 || d->heads < 1 || d->heads > 2 || !eprecon_mlp4x_supported(d->channels, d->out_channels)) return EPRECON_ERR_ARG;
    if (d->n == 0) return EPRECON_OK;
    if (!d->x || d->ld_x < d->channels || d->n > 0x7fffffffll - 64) return EPRECON_ERR_ARG;
    if (d->residual && d->channels != d->out_channels) return EPRECON_ERR_ARG;
    MlpParams p;
    p.x = d->x; p.ldx = (int)d->ld_x; p.n = (int)d->n; p.C = d->channels; p.Cout = d->out_channels; p.residual = d->residual;
    p.vec4 = (d->ld_x % 4 == 0) && ((reinterpret_cast<uintptr_t>(d->x) & 15) == 0);
    p.eps1 = d->eps1; p.eps2 = d->eps2;
    for (int h = 0; h < 2; ++h) {
        const eprecon_mlp4x_head &s = d->head[h < d->heads ? h : 0];
        if (!s.w1 || !s.b1 || !s.g1 || !s.be1 || !s.w2 || !s.b2 || !s.g2 || !s.be2 || !s.w3 || !s.b3 || !s.y || s.ld_y < d->out_channels)
            return EPRECON_ERR_ARG;
        const float *ptrs[] = {s.w1, s.b1, s.g1, s.be1, s.w2, s.b2, s.g2, s.be2, s.w3, s.b3};
        for (const float *q : ptrs)
            if (reinterpret_cast<uintptr_t>(q) & 15) return EPRECON_ERR_ARG;
        p.w1[h] = s.w1; p.b1[h] = s.b1; p.g1[h] = s.g1; p.be1[h] = s.be1;
        p.w2[h] = s.w2; p.b2[h] = s.b2; p.g2[h] = s.g2; p.be2[h] = s.be2;
        p.w3[h] = s.w3; p.b3[h] = s.b3; p.y[h] = s.y; p.ldy[h] = (int)s.ld_y;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool proj = d->out_channels > 16;
    switch (d->channels) {
        case 24: return launch<2, 6, 1>(p, d->heads, st);
        case 96: return launch<6, 24, 1>(p, d->heads, st);
        case 48: return proj ? launch<3, 12, 3>(p, d->heads, st) : launch<3, 12, 1>(p, d->heads, st);
        case 88: return launch<6, 22, 3>(p, d->heads, st);
        case 176: return launch<11, 44, 3>(p, d->heads, st);
    }
    return EPRECON_ERR_UNSUPPORTED;
}

}  // extern "C"
