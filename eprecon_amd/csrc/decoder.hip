// Voxel side of the mask-transformer decoder on gfx950: everything of MultiScaleMaskedTransformerDecoder.forward
// (models/mask3dformer.py:337-445 of the reference) whose shape depends on the number of voxels of a level.
//
//   decoder_keys_kernel               src = feats + level_embed, keys = src + Fourier position encoding of the voxel
//                                     coordinates (models/voxel_position_encoding.py:123-152, models/mask3dformer.py:346-357):
//                                     one launch per level instead of ~12 elementwise / GEMM launches
//   masked_attention_partial_kernel   the masked cross-attention of a decoder layer (nn.MultiheadAttention with
//   masked_attention_reduce_kernel    attn_mask = sigmoid(mask logits at the level's voxels) < 0.5, a query whose mask blocks
//                                     every voxel attends to all of them: models/mask3dformer.py:383-397,441-443) as a
//                                     split-K flash attention over the voxels: the [Q, N] attention mask, the [H, Q, N]
//                                     score / probability tensors and the index_select of the mask logits never exist.
//
// Layout: keys / values are voxel rows f32[N, C] (C = H * D), the mask logits arrive TRANSPOSED, f32[N_fine, Q] (one GEMM
// mask_features[N_fine, C] x mask_embed^T[C, Q]), so that the logits of the fine voxel a coarser voxel maps to
// (mask_rows[n], the cdist + argmin of models/mask3dformer.py:361-367) are one contiguous 4 Q-byte run.
// A workgroup owns a contiguous range of keys and Q * H / 2 threads: thread (q, hp) carries the online-softmax state of
// heads 2 hp and 2 hp + 1 of query q, twice — over the allowed keys and over all keys (the fallback of an all-blocked query,
// chosen in the reduce kernel by the exact count of allowed keys).  K / V tiles of 64 keys are staged in LDS and read as
// broadcasts; scores are rescaled once per 8 keys.  Partial states are merged in workgroup order: deterministic.
// HBM-bound by the contract (8 C bytes per key), VALU-bound in practice (~45 fp32 operations + 2 v_exp_f32 per key, head and
// query).  The softmax weights use v_exp_f32 (__expf, relative error ~1e-6); the mask decision uses torch.sigmoid's expression.
#include <math.h>

#include "common.hpp"

namespace {
using namespace ep;

constexpr int kAttTile = 64;   // keys per LDS tile
constexpr int kAttSub = 8;     // keys per register sub-tile (one rescale of the running state per sub-tile)
constexpr float kNegBig = -1.0e30f;

__global__ __launch_bounds__(256) void decoder_keys_kernel(const int32_t *coords, int ld_c, const float *feats, int ld_f,
                                                           const float *level_embed, const float *gauss_b, float hx, float hy,
                                                           float hz, int n, int C, float *src, float *keys)
{
    const int half = C / 2;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * half) return;
    const int i = (int)(e / half), c = (int)(e - (int64_t)i * half);
    // (x - lo) * 1.0 / (hi - lo) + 0.0 with lo = 0, then * 2 pi, then [N,3] x [3, C/2] as a k-ordered fma chain
    const float two_pi = 6.283185307179586f;
    const float x = ((float)coords[(size_t)i * ld_c + 0] / hx) * two_pi;
    const float y = ((float)coords[(size_t)i * ld_c + 1] / hy) * two_pi;
    const float z = ((float)coords[(size_t)i * ld_c + 2] / hz) * two_pi;
    float p = x * gauss_b[c];
    p = fmaf(y, gauss_b[half + c], p);
    p = fmaf(z, gauss_b[2 * half + c], p);
    const float s0 = feats[(size_t)i * ld_f + c] + level_embed[c];
    const float s1 = feats[(size_t)i * ld_f + half + c] + level_embed[half + c];
    src[(size_t)i * C + c] = s0;
    src[(size_t)i * C + half + c] = s1;
    keys[(size_t)i * C + c] = s0 + sinf(p);
    keys[(size_t)i * C + half + c] = s1 + cosf(p);
}

struct AttParams {
    const float *q;         // element (h, i, d) at q[h * q_sh + i * q_sq + d]
    int q_sh, q_sq;
    const float *k, *v;     // [N][ld]
    int ld_k, ld_v;
    const float *logits_t;  // [N_fine][ld_l] or nullptr (no mask)
    int ld_l;
    const int32_t *rows;    // [N] row of logits_t per key, nullptr = identity
    int n_fine;
    int N, Q, H;
    float scale;
    int keys_per_wg;
    float *partial;         // [G][2][H][Q][D + 2]: (running max, sum, weighted values); [.][0] allowed keys, [.][1] all keys
    int32_t *allowed;       // [G][Q] allowed keys of this workgroup's range
};

template <int D>
__global__ __launch_bounds__(512) void masked_attention_partial_kernel(AttParams p)
{
    static_assert((2 * D) % 4 == 0, "a head pair is read as float4s");
    constexpr int F4 = 2 * D / 4;   // float4s per head pair
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = p.H * D;
    float *sK = reinterpret_cast<float *>(smem);              // [kAttTile][C]
    float *sV = sK + kAttTile * C;                            // [kAttTile][C]
    unsigned char *sBlk = reinterpret_cast<unsigned char *>(sV + kAttTile * C);   // [kAttTile][Q]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int qi = tid % p.Q, hp = tid / p.Q;                 // this thread: query qi, heads 2 hp and 2 hp + 1
    const int k_begin = (int)blockIdx.x * p.keys_per_wg;
    const int k_end = min(k_begin + p.keys_per_wg, p.N);

    float qa[D], qb[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        qa[d] = p.q[(size_t)(2 * hp) * p.q_sh + (size_t)qi * p.q_sq + d] * p.scale;
        qb[d] = p.q[(size_t)(2 * hp + 1) * p.q_sh + (size_t)qi * p.q_sq + d] * p.scale;
    }
    // running states: [variant 0 = allowed keys, 1 = all keys][head a / b]
    float m[2][2], l[2][2], o[2][2][D];
#pragma unroll
    for (int var = 0; var < 2; ++var)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            m[var][h] = kNegBig;
            l[var][h] = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) o[var][h][d] = 0.0f;
        }
    int n_allowed = 0;

    for (int t0 = k_begin; t0 < k_end; t0 += kAttTile) {
        const int tn = min(kAttTile, k_end - t0);
        __syncthreads();   // the previous tile is consumed
        const int c4 = C / 4;
        for (int e = tid; e < tn * c4; e += nthr) {
            const int t = e / c4, c = e - t * c4;
            reinterpret_cast<float4 *>(sK)[t * c4 + c] = *reinterpret_cast<const float4 *>(p.k + (size_t)(t0 + t) * p.ld_k + 4 * c);
            reinterpret_cast<float4 *>(sV)[t * c4 + c] = *reinterpret_cast<const float4 *>(p.v + (size_t)(t0 + t) * p.ld_v + 4 * c);
        }
        for (int e = tid; e < tn * p.Q; e += nthr) {
            const int t = e / p.Q, qq = e - t * p.Q;
            unsigned char blocked = 0;
            if (p.logits_t) {
                const int row = p.rows ? p.rows[t0 + t] : t0 + t;
                blocked = 1;
                if (row >= 0 && row < p.n_fine) {
                    const float x = p.logits_t[(size_t)row * p.ld_l + qq];
                    blocked = (1.0f / (1.0f + expf(-x))) < 0.5f ? 1 : 0;   // torch.sigmoid's expression, then `< 0.5`
                }
            }
            sBlk[t * p.Q + qq] = blocked;
        }
        __syncthreads();

        for (int s0 = 0; s0 < tn; s0 += kAttSub) {
            float sa[kAttSub], sb[kAttSub];
            unsigned okmask = 0, alwmask = 0;      // bit j: key s0 + j exists / exists and is allowed for this query
            float mx[2][2] = {{kNegBig, kNegBig}, {kNegBig, kNegBig}};
#pragma unroll
            for (int j = 0; j < kAttSub; ++j) {
                const int t = s0 + j;
                const bool ok = t < tn;
                const int tt = ok ? t : tn - 1;
                const float4 *kr = reinterpret_cast<const float4 *>(sK + tt * C + 2 * D * hp);
                float kv[2 * D];
#pragma unroll
                for (int f = 0; f < F4; ++f) {
                    const float4 x = kr[f];
                    kv[4 * f] = x.x; kv[4 * f + 1] = x.y; kv[4 * f + 2] = x.z; kv[4 * f + 3] = x.w;
                }
                float a = 0.0f, b = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    a = fmaf(qa[d], kv[d], a);
                    b = fmaf(qb[d], kv[D + d], b);
                }
                const bool alw = ok && sBlk[tt * p.Q + qi] == 0;
                sa[j] = a; sb[j] = b;
                okmask |= (ok ? 1u : 0u) << j;
                alwmask |= (alw ? 1u : 0u) << j;
                if (ok) { mx[1][0] = fmaxf(mx[1][0], a); mx[1][1] = fmaxf(mx[1][1], b); }
                if (alw) { mx[0][0] = fmaxf(mx[0][0], a); mx[0][1] = fmaxf(mx[0][1], b); }
            }
            if (hp == 0) n_allowed += __popc(alwmask);
            // one rescale of the running states per sub-tile
#pragma unroll
            for (int var = 0; var < 2; ++var)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float m_new = fmaxf(m[var][h], mx[var][h]);
                    const float r = __expf(m[var][h] - m_new);   // (both kNegBig: exp(0) = 1 on an all-zero state)
                    l[var][h] *= r;
#pragma unroll
                    for (int d = 0; d < D; ++d) o[var][h][d] *= r;
                    m[var][h] = m_new;
                }
#pragma unroll
            for (int j = 0; j < kAttSub; ++j) {
                if (!((okmask >> j) & 1u)) continue;
                const int t = s0 + j;
                const float4 *vr = reinterpret_cast<const float4 *>(sV + t * C + 2 * D * hp);
                float vv[2 * D];
#pragma unroll
                for (int f = 0; f < F4; ++f) {
                    const float4 x = vr[f];
                    vv[4 * f] = x.x; vv[4 * f + 1] = x.y; vv[4 * f + 2] = x.z; vv[4 * f + 3] = x.w;
                }
                const float pa = __expf(sa[j] - m[1][0]), pb = __expf(sb[j] - m[1][1]);
                l[1][0] += pa; l[1][1] += pb;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    o[1][0][d] = fmaf(pa, vv[d], o[1][0][d]);
                    o[1][1][d] = fmaf(pb, vv[D + d], o[1][1][d]);
                }
                if ((alwmask >> j) & 1u) {
                    const float ma = __expf(sa[j] - m[0][0]), mb = __expf(sb[j] - m[0][1]);
                    l[0][0] += ma; l[0][1] += mb;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        o[0][0][d] = fmaf(ma, vv[d], o[0][0][d]);
                        o[0][1][d] = fmaf(mb, vv[D + d], o[0][1][d]);
                    }
                }
            }
        }
    }
    constexpr int W = D + 2;
#pragma unroll
    for (int var = 0; var < 2; ++var)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float *dst = p.partial + ((((size_t)blockIdx.x * 2 + var) * p.H + (2 * hp + h)) * p.Q + qi) * W;
            dst[0] = m[var][h];
            dst[1] = l[var][h];
#pragma unroll
            for (int d = 0; d < D; ++d) dst[2 + d] = o[var][h][d];
        }
    if (hp == 0) p.allowed[(size_t)blockIdx.x * p.Q + qi] = n_allowed;
}

// one wave per (head, query): lane j merges the partial states of workgroups j, j + 64, ... in order, the 64 lane states
// are merged by a fixed xor-butterfly (deterministic); a query without an allowed key takes the all-keys state
// (models/mask3dformer.py:388).  out[h][q][d]: what scaled_dot_product_attention returns per head.
template <int D>
__global__ __launch_bounds__(64) void masked_attention_reduce_kernel(const float *partial, const int32_t *allowed, int G, int Q,
                                                                     int H, int use_mask, float *out)
{
    constexpr int W = D + 2;
    const int e = blockIdx.x, lane = threadIdx.x;
    const int h = e / Q, q = e - h * Q;
    int var = 1;
    if (use_mask) {
        int cnt = 0;
        for (int g = lane; g < G; g += 64) cnt += allowed[(size_t)g * Q + q] > 0 ? 1 : 0;
#pragma unroll
        for (int msk = 32; msk > 0; msk >>= 1) cnt += __shfl_xor(cnt, msk);
        var = cnt > 0 ? 0 : 1;
    }
    float M = kNegBig, L = 0.0f, O[D];
#pragma unroll
    for (int d = 0; d < D; ++d) O[d] = 0.0f;
    for (int g = lane; g < G; g += 64) {
        const float4 *src = reinterpret_cast<const float4 *>(partial + ((((size_t)g * 2 + var) * H + h) * Q + q) * W);
        const float4 s0 = src[0], s1 = src[1];            // (m, l, o0, o1), (o2, o3, o4, o5)
        if (s0.y == 0.0f) continue;                       // an empty partial state (its max is the sentinel)
        const float m_new = fmaxf(M, s0.x);
        const float ra = __expf(M - m_new), rb = __expf(s0.x - m_new);
        L = L * ra + s0.y * rb;
        O[0] = O[0] * ra + s0.z * rb; O[1] = O[1] * ra + s0.w * rb;
        O[2] = O[2] * ra + s1.x * rb; O[3] = O[3] * ra + s1.y * rb;
        O[4] = O[4] * ra + s1.z * rb; O[5] = O[5] * ra + s1.w * rb;
        M = m_new;
    }
#pragma unroll
    for (int msk = 32; msk > 0; msk >>= 1) {
        const float oM = __shfl_xor(M, msk), oL = __shfl_xor(L, msk);
        float oO[D];
#pragma unroll
        for (int d = 0; d < D; ++d) oO[d] = __shfl_xor(O[d], msk);
        const float m_new = fmaxf(M, oM);
        const float ra = __expf(M - m_new), rb = __expf(oM - m_new);
        L = L * ra + oL * rb;
#pragma unroll
        for (int d = 0; d < D; ++d) O[d] = O[d] * ra + oO[d] * rb;
        M = m_new;
    }
    if (lane == 0) {
        const float inv = L > 0.0f ? 1.0f / L : 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) out[((size_t)h * Q + q) * D + d] = O[d] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Query side of a decoder layer (static shape [Q, C]): everything between two cross-attentions — out-projection + residual +
// LayerNorm of the cross-attention, self-attention over the Q queries, FFN, the prediction head's class / mask embeddings and
// the next layer's projected queries (models/mask3dformer.py:399-445 with the post-norm blocks of :33-196).  ~25 PyTorch
// launches per layer (7 GEMMs of [80, 48] x [48, 48..192], LayerNorms, an 80 x 80 attention, ...) become TWO: every step but
// the self-attention is row-wise, so a workgroup owns kQsRows query rows and walks the layer as vector-matrix products — the
// row in LDS, the weights pre-transposed to [in][out] so that thread n reads column n coalesced (they stay in L2: 350 KB per
// layer, read by every workgroup); the self-attention needs every row's keys / values and is the one grid-wide dependency:
//   query_side_a:  u = o W_o + b; t1 = LN(x + u); [Qs | Ks] = (t1 + pos) W_qk + b; Vs = t1 W_v + b      -> workspace
//   query_side_b:  A = softmax(Qs Ks^T / sqrt(d)) Vs per head; t2 = LN(t1 + A W_o' + b); t3 = LN(t2 + FFN(t2));
//                  dec = LN_dec(t3); class logits, mask embedding MLP, next layer's q = (t3 + pos) W_q' + b
// Sums run in a fixed order (k ascending inside a part, parts in order): deterministic; equal to the PyTorch modules within
// fp32 round-off.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kQsRows = 2;        // query rows per workgroup
constexpr int kQsThreads = 384;   // >= the widest layer (ffn_dim, mask_hidden); the narrow layers split their sum over 384 / N parts
constexpr int kQsMaxC = 64;       // channels (one wave holds a row for the LayerNorms)

// y[r][n] = [relu]( bias[n] + sum_k x[r][k] * wt[k][n] ) for the workgroup's rows; x / y in LDS (row pitch ldx / ldy), wt [K][N]
// (row pitch ldw) in global memory.  The threads split into parts = blockDim / N groups of N; part p takes a contiguous range of k.
__device__ __forceinline__ void qs_gemv(const float *__restrict__ wt, int ldw, const float *__restrict__ bias, const float *sx, int ldx,
                                        float *sy, int ldy, int K, int N, bool relu, float *sPart)
{
    const int tid = threadIdx.x;
    const int parts = max(1, min((int)blockDim.x / N, K));
    const int part = tid / N, n = tid - part * N;
    const int kchunk = (K + parts - 1) / parts;
    float acc[kQsRows];
#pragma unroll
    for (int r = 0; r < kQsRows; ++r) acc[r] = 0.0f;
    if (part < parts) {
        const int k0 = part * kchunk, k1 = min(K, k0 + kchunk);
        // eight weights in flight per thread (the chain of L2 latencies, not the arithmetic, is what a layer costs); the sum
        // itself runs k-ascending as before
        constexpr int U = 8;
        for (int k = k0; k < k1; k += U) {
            float w[U];
#pragma unroll
            for (int i = 0; i < U; ++i) w[i] = k + i < k1 ? wt[(size_t)(k + i) * ldw + n] : 0.0f;
#pragma unroll
            for (int i = 0; i < U; ++i)
                if (k + i < k1) {
#pragma unroll
                    for (int r = 0; r < kQsRows; ++r) acc[r] = fmaf(sx[r * ldx + k + i], w[i], acc[r]);
                }
        }
        if (parts > 1) {
#pragma unroll
            for (int r = 0; r < kQsRows; ++r) sPart[(r * parts + part) * N + n] = acc[r];
        }
    }
    if (parts > 1) __syncthreads();
    if (part == 0) {
        const float b = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < kQsRows; ++r) {
            float v = acc[r];
            for (int p = 1; p < parts; ++p) v += sPart[(r * parts + p) * N + n];
            v += b;
            sy[r * ldy + n] = relu ? fmaxf(v, 0.0f) : v;
        }
    }
    __syncthreads();
}

// sy[r][:] = LayerNorm(sa[r][:] (+ sb[r][:])) * g + b over C <= 64 channels: wave r owns row r
__device__ __forceinline__ void qs_layernorm(const float *sa, const float *sb, int ld, float *sy, int ldy, int C, const float *g,
                                             const float *b, float eps)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < kQsRows) {
        float v = 0.0f;
        if (lane < C) v = sa[wave * ld + lane] + (sb ? sb[wave * ld + lane] : 0.0f);
        float s = v;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
        const float mean = s / (float)C;
        const float d = lane < C ? v - mean : 0.0f;
        float q = d * d;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) q += __shfl_xor(q, m);
        const float inv = 1.0f / sqrtf(q / (float)C + eps);
        if (lane < C) sy[wave * ldy + lane] = d * inv * g[lane] + b[lane];
    }
    __syncthreads();
}

struct QsParams {
    int Q, C, H, FF, K1, MH;
    const float *o_attn, *state_in, *query_pos;
    const float *cross_out_wt, *cross_out_b, *cross_ln_g, *cross_ln_b;
    const float *self_in_wt, *self_in_b, *self_out_wt, *self_out_b, *self_ln_g, *self_ln_b;
    const float *ffn1_wt, *ffn1_b, *ffn2_wt, *ffn2_b, *ffn_ln_g, *ffn_ln_b;
    const float *dec_ln_g, *dec_ln_b, *cls_wt, *cls_b, *m1_wt, *m1_b, *m2_wt, *m2_b, *m3_wt, *m3_b;
    const float *next_q_wt, *next_q_b;
    float eps;
    float *state_out, *cls_out, *me_out, *next_q_out;
    float *ws;   // [4][Q][C]: t1, Qs, Ks, Vs
};

__global__ __launch_bounds__(kQsThreads) void query_side_a_kernel(QsParams p)
{
    __shared__ float sX[kQsRows * kQsMaxC], sO[kQsRows * kQsMaxC], sU[kQsRows * kQsMaxC], sT[kQsRows * kQsMaxC];
    __shared__ float sY[kQsRows * 3 * kQsMaxC], sPart[kQsThreads * kQsRows];
    const int tid = threadIdx.x, C = p.C, D = C / p.H;
    const int q0 = blockIdx.x * kQsRows;
    for (int e = tid; e < kQsRows * C; e += kQsThreads) {
        const int r = e / C, c = e - r * C;
        const int q = min(q0 + r, p.Q - 1);
        sX[r * kQsMaxC + c] = p.state_in[(size_t)q * C + c];
        sO[r * kQsMaxC + c] = p.o_attn[((size_t)(c / D) * p.Q + q) * D + (c % D)];   // [H][Q][D] -> row q, heads side by side
    }
    __syncthreads();
    qs_gemv(p.cross_out_wt, C, p.cross_out_b, sO, kQsMaxC, sU, kQsMaxC, C, C, false, sPart);
    qs_layernorm(sX, sU, kQsMaxC, sT, kQsMaxC, C, p.cross_ln_g, p.cross_ln_b, p.eps);
    for (int e = tid; e < kQsRows * C; e += kQsThreads) {
        const int r = e / C, c = e - r * C;
        sX[r * kQsMaxC + c] = sT[r * kQsMaxC + c] + p.query_pos[(size_t)min(q0 + r, p.Q - 1) * C + c];
    }
    __syncthreads();
    // the packed in-projection [C][3C]: columns [0, 2C) = q | k (from t1 + pos), columns [2C, 3C) = v (from t1)
    qs_gemv(p.self_in_wt, 3 * C, p.self_in_b, sX, kQsMaxC, sY, 3 * kQsMaxC, C, 2 * C, false, sPart);
    qs_gemv(p.self_in_wt + 2 * C, 3 * C, p.self_in_b + 2 * C, sT, kQsMaxC, sY + 2 * C, 3 * kQsMaxC, C, C, false, sPart);
    for (int e = tid; e < kQsRows * C; e += kQsThreads) {
        const int r = e / C, c = e - r * C;
        const int q = q0 + r;
        if (q >= p.Q) continue;
        p.ws[((size_t)0 * p.Q + q) * C + c] = sT[r * kQsMaxC + c];
        p.ws[((size_t)1 * p.Q + q) * C + c] = sY[r * 3 * kQsMaxC + c];
        p.ws[((size_t)2 * p.Q + q) * C + c] = sY[r * 3 * kQsMaxC + C + c];
        p.ws[((size_t)3 * p.Q + q) * C + c] = sY[r * 3 * kQsMaxC + 2 * C + c];
    }
}

constexpr int kQsMaxQ = 128;      // queries (the self-attention's key axis, held in LDS per head)
constexpr int kQsMaxW = 256;      // widest hidden layer
__global__ __launch_bounds__(kQsThreads) void query_side_b_kernel(QsParams p)
{
    __shared__ float sT1[kQsRows * kQsMaxC], sA[kQsRows * kQsMaxC], sU[kQsRows * kQsMaxC], sT2[kQsRows * kQsMaxC];
    __shared__ float sT3[kQsRows * kQsMaxC], sDec[kQsRows * kQsMaxC], sQ[kQsRows * kQsMaxC];
    __shared__ float sH1[kQsRows * kQsMaxW], sH2[kQsRows * kQsMaxW], sPart[kQsThreads * kQsRows];
    __shared__ float sS[kQsRows * 8 * kQsMaxQ];     // scores / probabilities [row][head][key]  (H <= 8)
    __shared__ float sKV[kQsMaxQ * kQsMaxC];        // every query's self-attention keys, then values ([Q][C], 20 KB at 80 x 48)
    const int tid = threadIdx.x, C = p.C, H = p.H, D = C / H, Q = p.Q;
    const int q0 = blockIdx.x * kQsRows;
    const float *t1 = p.ws, *Qs = p.ws + (size_t)Q * C, *Ks = p.ws + (size_t)2 * Q * C, *Vs = p.ws + (size_t)3 * Q * C;
    for (int e = tid; e < kQsRows * C; e += kQsThreads) {
        const int r = e / C, c = e - r * C;
        const int q = min(q0 + r, Q - 1);
        sT1[r * kQsMaxC + c] = t1[(size_t)q * C + c];
        sQ[r * kQsMaxC + c] = Qs[(size_t)q * C + c];
    }
    for (int e = tid; e < Q * C; e += kQsThreads) sKV[e] = Ks[e];
    __syncthreads();
    // ---- self-attention of the workgroup's rows over all Q keys ----
    const float scale = 1.0f / sqrtf((float)D);
    for (int e = tid; e < kQsRows * H * Q; e += kQsThreads) {
        const int r = e / (H * Q), h = (e / Q) % H, j = e % Q;
        float sdot = 0.0f;
        for (int d = 0; d < D; ++d) sdot = fmaf(sQ[r * kQsMaxC + h * D + d] * scale, sKV[j * C + h * D + d], sdot);
        sS[(r * H + h) * kQsMaxQ + j] = sdot;
    }
    __syncthreads();
    for (int e = tid; e < Q * C; e += kQsThreads) sKV[e] = Vs[e];   // (published by the barrier behind the softmax)
    {   // softmax over the keys: one wave per (row, head) pair in turn
        const int wave = tid >> 6, lane = tid & 63, nw = kQsThreads / 64;
        for (int rh = wave; rh < kQsRows * H; rh += nw) {
            float *row = sS + rh * kQsMaxQ;
            float mx = -3.0e38f;
            for (int j = lane; j < Q; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
            float sum = 0.0f;
            for (int j = lane; j < Q; j += 64) {
                const float e_ = expf(row[j] - mx);
                row[j] = e_;
                sum += e_;
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
            const float inv = 1.0f / sum;
            for (int j = lane; j < Q; j += 64) row[j] *= inv;
        }
    }
    __syncthreads();
    for (int e = tid; e < kQsRows * C; e += kQsThreads) {
        const int r = e / C, c = e - r * C, h = c / D;
        const float *pr = sS + (r * H + h) * kQsMaxQ;
        float a = 0.0f;
        for (int j = 0; j < Q; ++j) a = fmaf(pr[j], sKV[j * C + c], a);
        sA[r * kQsMaxC + c] = a;
    }
    __syncthreads();
    qs_gemv(p.self_out_wt, C, p.self_out_b, sA, kQsMaxC, sU, kQsMaxC, C, C, false, sPart);
    qs_layernorm(sT1, sU, kQsMaxC, sT2, kQsMaxC, C, p.self_ln_g, p.self_ln_b, p.eps);
    // ---- FFN ----
    qs_gemv(p.ffn1_wt, p.FF, p.ffn1_b, sT2, kQsMaxC, sH1, kQsMaxW, C, p.FF, true, sPart);
    qs_gemv(p.ffn2_wt, C, p.ffn2_b, sH1, kQsMaxW, sU, kQsMaxC, p.FF, C, false, sPart);
    qs_layernorm(sT2, sU, kQsMaxC, sT3, kQsMaxC, C, p.ffn_ln_g, p.ffn_ln_b, p.eps);
    // ---- prediction head + the next layer's queries ----
    qs_layernorm(sT3, nullptr, kQsMaxC, sDec, kQsMaxC, C, p.dec_ln_g, p.dec_ln_b, p.eps);
    qs_gemv(p.cls_wt, p.K1, p.cls_b, sDec, kQsMaxC, sU, kQsMaxC, C, p.K1, false, sPart);
    for (int e = tid; e < kQsRows * p.K1; e += kQsThreads) {
        const int r = e / p.K1, c = e - r * p.K1;
        if (q0 + r < Q) p.cls_out[(size_t)(q0 + r) * p.K1 + c] = sU[r * kQsMaxC + c];
    }
    __syncthreads();
    qs_gemv(p.m1_wt, p.MH, p.m1_b, sDec, kQsMaxC, sH1, kQsMaxW, C, p.MH, true, sPart);
    qs_gemv(p.m2_wt, p.MH, p.m2_b, sH1, kQsMaxW, sH2, kQsMaxW, p.MH, p.MH, true, sPart);
    qs_gemv(p.m3_wt, C, p.m3_b, sH2, kQsMaxW, sU, kQsMaxC, p.MH, C, false, sPart);
    for (int e = tid; e < kQsRows * C; e += kQsThreads) {
        const int r = e / C, c = e - r * C;
        const int q = q0 + r;
        if (q < Q) {
            p.me_out[(size_t)q * C + c] = sU[r * kQsMaxC + c];
            p.state_out[(size_t)q * C + c] = sT3[r * kQsMaxC + c];
        }
        sA[r * kQsMaxC + c] = sT3[r * kQsMaxC + c] + p.query_pos[(size_t)min(q, Q - 1) * C + c];
    }
    __syncthreads();
    if (p.next_q_wt) {
        qs_gemv(p.next_q_wt, C, p.next_q_b, sA, kQsMaxC, sU, kQsMaxC, C, C, false, sPart);
        for (int e = tid; e < kQsRows * C; e += kQsThreads) {
            const int r = e / C, c = e - r * C;
            if (q0 + r < Q) p.next_q_out[(size_t)(q0 + r) * C + c] = sU[r * kQsMaxC + c];
        }
    }
}

int att_groups(int64_t n_keys, int *keys_per_wg)
{
    // ~3 workgroups per CU; ranges are whole LDS tiles
    const int64_t target = 768;
    int64_t per = ceil_div(ceil_div(n_keys, target), (int64_t)kAttTile) * kAttTile;
    if (per < kAttTile) per = kAttTile;
    *keys_per_wg = (int)per;
    return (int)ceil_div(n_keys, per);
}

bool att_shape_ok(int n_queries, int n_heads, int head_dim)
{
    return head_dim == 6 && n_heads > 0 && n_heads % 2 == 0 && n_queries > 0 && n_queries * (n_heads / 2) <= 512 &&
           n_queries <= 255 && (n_heads * head_dim) % 4 == 0;
}

}  // namespace

extern "C" {

int eprecon_decoder_keys_async(const int32_t *coords, int ld_coords, const float *feats, int ld_feats, const float *level_embed,
                               const float *gauss_b, const float *extent_host, int64_t n, int channels, float *src_out,
                               float *keys_out, void *stream)
{
    if (n < 0 || channels <= 0 || channels % 2 || !extent_host || ld_coords < 3 || ld_feats < channels ||
        (n > 0 && (!coords || !feats || !level_embed || !gauss_b || !src_out || !keys_out)) || n * channels > 0x7fffffffll * 128)
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(decoder_keys_kernel, dim3((unsigned)ceil_div(n * (channels / 2), (int64_t)256)), dim3(256), 0,
                       (hipStream_t)stream, coords, ld_coords, feats, ld_feats, level_embed, gauss_b, extent_host[0], extent_host[1],
                       extent_host[2], (int)n, channels, src_out, keys_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

size_t eprecon_masked_attention_workspace_bytes(int64_t n_keys, int n_queries, int n_heads, int head_dim)
{
    if (n_keys <= 0 || !att_shape_ok(n_queries, n_heads, head_dim)) return 0;
    int per;
    const int G = att_groups(n_keys, &per);
    return align_up((size_t)G * 2 * n_heads * n_queries * (head_dim + 2) * sizeof(float), 256) +
           align_up((size_t)G * n_queries * sizeof(int32_t), 256);
}

int eprecon_masked_attention_async(const float *q, int q_stride_head, int q_stride_query, const float *k, int ld_k, const float *v, int ld_v, int64_t n_keys,
                                   const float *mask_logits_t, int ld_mask, const int32_t *mask_rows, int64_t n_mask_rows,
                                   int n_queries, int n_heads, int head_dim, float scale, float *out, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (!q || !k || !v || !out || n_keys <= 0 || n_keys > 0x7fffffff || n_mask_rows < 0 || n_mask_rows > 0x7fffffff ||
        q_stride_head <= 0 || q_stride_query <= 0)
        return EPRECON_ERR_ARG;
    if (!att_shape_ok(n_queries, n_heads, head_dim)) return EPRECON_ERR_UNSUPPORTED;
    const int C = n_heads * head_dim;
    if (ld_k < C || ld_v < C || ld_k % 4 || ld_v % 4 || (reinterpret_cast<uintptr_t>(k) & 15) || (reinterpret_cast<uintptr_t>(v) & 15))
        return EPRECON_ERR_UNSUPPORTED;
    if (mask_logits_t && (ld_mask < n_queries || n_mask_rows <= 0 || (!mask_rows && n_mask_rows < n_keys))) return EPRECON_ERR_ARG;
    if (!workspace || workspace_bytes < eprecon_masked_attention_workspace_bytes(n_keys, n_queries, n_heads, head_dim))
        return EPRECON_ERR_WORKSPACE;
    AttParams p;
    p.q = q; p.q_sh = q_stride_head; p.q_sq = q_stride_query; p.k = k; p.v = v; p.ld_k = ld_k; p.ld_v = ld_v;
    p.logits_t = mask_logits_t; p.ld_l = ld_mask; p.rows = mask_rows; p.n_fine = (int)n_mask_rows;
    p.N = (int)n_keys; p.Q = n_queries; p.H = n_heads; p.scale = scale;
    const int G = att_groups(n_keys, &p.keys_per_wg);
    char *ws = reinterpret_cast<char *>(workspace);
    p.partial = reinterpret_cast<float *>(ws);
    p.allowed = reinterpret_cast<int32_t *>(ws + align_up((size_t)G * 2 * n_heads * n_queries * (head_dim + 2) * sizeof(float), 256));
    const int threads = n_queries * (n_heads / 2);
    const size_t lds = (size_t)2 * kAttTile * C * sizeof(float) + (size_t)kAttTile * n_queries;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((masked_attention_partial_kernel<6>), dim3((unsigned)G), dim3((unsigned)threads), lds, st, p);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL((masked_attention_reduce_kernel<6>), dim3((unsigned)(n_heads * n_queries)), dim3(64), 0, st,
                       (const float *)p.partial, (const int32_t *)p.allowed, G, n_queries, n_heads, mask_logits_t ? 1 : 0, out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_decoder_query_side_async(const eprecon_decoder_layer_desc *d, void *stream)
{
    if (!d || !d->o_attn || !d->state_in || !d->query_pos || !d->workspace || !d->state_out || !d->cls_out || !d->mask_embed_out ||
        !d->cross_out_wt || !d->cross_out_b || !d->cross_ln_g || !d->cross_ln_b || !d->self_in_wt || !d->self_in_b || !d->self_out_wt ||
        !d->self_out_b || !d->self_ln_g || !d->self_ln_b || !d->ffn1_wt || !d->ffn1_b || !d->ffn2_wt || !d->ffn2_b || !d->ffn_ln_g ||
        !d->ffn_ln_b || !d->dec_ln_g || !d->dec_ln_b || !d->cls_wt || !d->cls_b || !d->m1_wt || !d->m1_b || !d->m2_wt || !d->m2_b ||
        !d->m3_wt || !d->m3_b || (d->next_q_wt && (!d->next_q_b || !d->next_q_out)))
        return EPRECON_ERR_ARG;
    const int Q = d->n_queries, C = d->channels, H = d->n_heads;
    if (Q <= 0 || Q > kQsMaxQ || C <= 0 || C > kQsMaxC || H <= 0 || H > 8 || C % H || d->ffn_dim <= 0 || d->ffn_dim > kQsThreads ||
        d->mask_hidden <= 0 || d->mask_hidden > kQsThreads || d->n_class_logits <= 0 || d->n_class_logits > kQsMaxC || 3 * C > kQsThreads)
        return EPRECON_ERR_UNSUPPORTED;
    QsParams p;
    p.Q = Q; p.C = C; p.H = H; p.FF = d->ffn_dim; p.K1 = d->n_class_logits; p.MH = d->mask_hidden;
    p.o_attn = d->o_attn; p.state_in = d->state_in; p.query_pos = d->query_pos;
    p.cross_out_wt = d->cross_out_wt; p.cross_out_b = d->cross_out_b; p.cross_ln_g = d->cross_ln_g; p.cross_ln_b = d->cross_ln_b;
    p.self_in_wt = d->self_in_wt; p.self_in_b = d->self_in_b; p.self_out_wt = d->self_out_wt; p.self_out_b = d->self_out_b;
    p.self_ln_g = d->self_ln_g; p.self_ln_b = d->self_ln_b;
    p.ffn1_wt = d->ffn1_wt; p.ffn1_b = d->ffn1_b; p.ffn2_wt = d->ffn2_wt; p.ffn2_b = d->ffn2_b; p.ffn_ln_g = d->ffn_ln_g; p.ffn_ln_b = d->ffn_ln_b;
    p.dec_ln_g = d->dec_ln_g; p.dec_ln_b = d->dec_ln_b; p.cls_wt = d->cls_wt; p.cls_b = d->cls_b;
    p.m1_wt = d->m1_wt; p.m1_b = d->m1_b; p.m2_wt = d->m2_wt; p.m2_b = d->m2_b; p.m3_wt = d->m3_wt; p.m3_b = d->m3_b;
    p.next_q_wt = d->next_q_wt; p.next_q_b = d->next_q_b;
    p.eps = d->ln_eps;
    p.state_out = d->state_out; p.cls_out = d->cls_out; p.me_out = d->mask_embed_out; p.next_q_out = d->next_q_out;
    p.ws = d->workspace;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div(Q, kQsRows);
    hipLaunchKernelGGL(query_side_a_kernel, dim3(grid), dim3(kQsThreads), 0, st, p);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(query_side_b_kernel, dim3(grid), dim3(kQsThreads), 0, st, p);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// panoptic_inference on the voxel side (models/mask3dformer.py:515-581): per voxel the kept query with the largest
// score * sigmoid(mask logit) (first one on ties) owns it; per query the number of owned voxels, of voxels with
// sigmoid >= 0.5, and of both.  ONE pass over the [Q, N] logits (coalesced along the voxels) instead of ~15 [Q, N] tensor
// ops, integer atomics only (LDS histogram per workgroup, then one global add per query): deterministic.  The host then
// decides per query (area ratio, stuff classes merged) and a second launch writes the segment ids.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kPanMaxQ = 256;

__global__ __launch_bounds__(256) void panoptic_stats_kernel(const float *logits, long long ld, const float *scores, const int *keep,
                                                             int Q, int N, int *owner, unsigned char *conf, int *counts)
{
    __shared__ int sCnt[3 * kPanMaxQ];
    __shared__ float sScore[kPanMaxQ];
    __shared__ int sKeep[kPanMaxQ];
    for (int i = threadIdx.x; i < 3 * Q; i += 256) sCnt[i] = 0;
    for (int i = threadIdx.x; i < Q; i += 256) { sScore[i] = scores[i]; sKeep[i] = keep[i]; }
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < N) {
        float best = -1.0f;
        int bq = -1, bconf = 0;
        for (int q = 0; q < Q; ++q) {
            if (!sKeep[q]) continue;                     // (uniform)
            const float x = logits[(size_t)q * ld + v];
            const float p = 1.0f / (1.0f + expf(-x));    // torch.sigmoid in fp32
            const float w = sScore[q] * p;
            const int c = p >= 0.5f;
            if (c) atomicAdd(&sCnt[Q + q], 1);
            if (w > best) { best = w; bq = q; bconf = c; }
        }
        owner[v] = bq;
        conf[v] = (unsigned char)bconf;
        if (bq >= 0) {
            atomicAdd(&sCnt[bq], 1);
            if (bconf) atomicAdd(&sCnt[2 * Q + bq], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * Q; i += 256)
        if (sCnt[i]) atomicAdd(counts + i, sCnt[i]);
}

__global__ void panoptic_assign_kernel(const int *owner, const unsigned char *conf, const int *idmap, int N, int *seg)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int q = owner[v];
    seg[v] = (q >= 0 && conf[v]) ? idmap[q] : 0;
}
}  // namespace

extern "C" {

int eprecon_panoptic_stats_async(const float *mask_logits, int64_t ld, const float *scores, const int32_t *keep, int n_queries,
                                 int64_t n, int32_t *owner_out, uint8_t *confident_out, int32_t *counts_out, void *stream)
{
    if (n < 0 || n > 0x7fffffff || n_queries <= 0 || n_queries > kPanMaxQ || ld < n || !scores || !keep || !counts_out) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    EP_HIP_CHECK(hipMemsetAsync(counts_out, 0, (size_t)3 * n_queries * sizeof(int32_t), st));
    if (n == 0) return EPRECON_OK;
    if (!mask_logits || !owner_out || !confident_out) return EPRECON_ERR_ARG;
    hipLaunchKernelGGL(panoptic_stats_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, st, mask_logits, (long long)ld, scores,
                       keep, n_queries, (int)n, owner_out, confident_out, counts_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_panoptic_assign_async(const int32_t *owner, const uint8_t *confident, const int32_t *idmap, int64_t n, int32_t *seg_out,
                                  void *stream)
{
    if (n < 0 || n > 0x7fffffff) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    if (!owner || !confident || !idmap || !seg_out) return EPRECON_ERR_ARG;
    hipLaunchKernelGGL(panoptic_assign_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, owner, confident,
                       idmap, (int)n, seg_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"
