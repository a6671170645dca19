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

}  // extern "C"
