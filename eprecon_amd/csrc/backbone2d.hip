// The 2D feeder's memory-bound layers on gfx950 (models/backbone.py:22-77 MnasMulti, run per view in TRAIN mode at test time:
// main.py:357, models/neuralrecon.py:53-54): depthwise k x k convolutions and the train-mode BatchNorm of a batch of views
// with SEPARATE statistics per view, on channels-last maps.  MIOpen serves the fp32 channels-last depthwise layers with its
// naive reference kernel (1.85 of the 4.4 ms of a 9-view pass, profiles/r04/backbone_kernels.txt) and PyTorch's instance-norm
// route costs four launches per BatchNorm; the point-wise (1 x 1) layers are plain GEMMs and stay on hipBLASLt.
//
//   bn_views_stats_kernel   a view's rows are cut into `chunks` ranges, one workgroup each: shifted sums -> (n, mean, M2) per channel
//   bn_views_finalize_kernel  the range summaries of a view merged in range order (Chan) -> the BatchNorm as
//                           scale = g / sqrt(var + eps), shift = b - mean * scale per (view, channel).  Deterministic.
//   bn_views_apply_kernel   y = [relu](x * scale + shift) [+ residual]   (in place or not)
//   dwconv_nhwc_kernel      out[n, oy, ox, c] = sum_taps w[tap][c] * f(x[n, oy S + dy - P, ox S + dx - P, c]) with zero padding;
//                           f = the PENDING BatchNorm + ReLU of the producer (scale / shift of the pixel's view) when given: the
//                           expanded maps (the largest tensors of the network) are then read once instead of read, written
//                           and read again.  A thread owns four channels of one output pixel; consecutive threads walk the
//                           channels, then the pixels of a row: every load / store is a coalesced 16-byte access.
#include "common.hpp"

namespace {
using namespace ep;

struct BnViewsParams {
    const float *x;
    int rows_per_view, C, chunks;
    float *partial;          // [V][chunks][3][C]
    const float *gamma, *beta;
    float eps;
    float *affine;           // [V][2][C]
};

__global__ __launch_bounds__(256) void bn_views_stats_kernel(BnViewsParams p)
{
    __shared__ float sS[1024], sQ[1024];
    const int tid = threadIdx.x, view = blockIdx.y, chunk = blockIdx.x;
    const int C = p.C, C4 = C >> 2, R = 256 / C4;
    const int c4 = tid % C4, r = tid / C4;
    const int per = (p.rows_per_view + p.chunks - 1) / p.chunks;
    const int r0 = chunk * per, r1 = min(p.rows_per_view, r0 + per);
    const float *base = p.x + (size_t)view * p.rows_per_view * C;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s, pv = s;
    if (r < R && r0 < r1) {
        pv = *reinterpret_cast<const float4 *>(base + (size_t)r0 * C + 4 * c4);     // pivot: the sums run on x - pivot
        // four rows in flight per thread (independent loads, one accumulator pair: the adds are cheap, the latency is not)
        int row = r0 + r;
        for (; row + 3 * R < r1; row += 4 * R) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(base + (size_t)(row + u * R) * C + 4 * c4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dx = v[u].x - pv.x, dy = v[u].y - pv.y, dz = v[u].z - pv.z, dw = v[u].w - pv.w;
                s.x += dx; s.y += dy; s.z += dz; s.w += dw;
                q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
            }
        }
        for (; row < r1; row += R) {
            const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)row * C + 4 * c4);
            const float dx = v.x - pv.x, dy = v.y - pv.y, dz = v.z - pv.z, dw = v.w - pv.w;
            s.x += dx; s.y += dy; s.z += dz; s.w += dw;
            q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
        }
    }
    if (r < R) {
        *reinterpret_cast<float4 *>(sS + r * C + 4 * c4) = s;
        *reinterpret_cast<float4 *>(sQ + r * C + 4 * c4) = q;
    }
    __syncthreads();
    float *dst = p.partial + ((size_t)view * p.chunks + chunk) * 3 * C;
    for (int c = tid; c < C; c += 256) {
        float ts = 0.0f, tq = 0.0f;
        for (int k = 0; k < R; ++k) { ts += sS[k * C + c]; tq += sQ[k * C + c]; }
        const float n = (float)max(r1 - r0, 0);
        const float pivot = r0 < r1 ? base[(size_t)r0 * C + c] : 0.0f;
        const float dm = n > 0.0f ? ts / n : 0.0f;
        dst[c] = n;
        dst[C + c] = pivot + dm;
        dst[2 * C + c] = fmaxf(tq - ts * dm, 0.0f);
    }
}

// (scale, shift) of a view from its range summaries: Chan's pairwise update in range order.  A second launch, not a "last
// workgroup merges" epilogue of the first: that needs a device-scope fence in every workgroup, and on a part whose eight XCDs
// have private L2s each such fence writes the L2 back — measured 40-50 us per BatchNorm against ~10 for the two launches.
__global__ __launch_bounds__(64) void bn_views_finalize_kernel(BnViewsParams p)
{
    const int tid = blockIdx.y * 64 + threadIdx.x, view = blockIdx.x, C = p.C;
    const float *src = p.partial + (size_t)view * p.chunks * 3 * C;
    for (int c = tid; c < C; c += 64 * (int)gridDim.y) {
        float n = 0.0f, mean = 0.0f, m2 = 0.0f;
        for (int k0 = 0; k0 < p.chunks; k0 += 16) {   // 48 loads in flight, then Chan's pairwise update in chunk order
            float nb[16], mb[16], qb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool ok = k0 + u < p.chunks;
                const float *e = src + (size_t)(ok ? k0 + u : 0) * 3 * C + c;
                nb[u] = ok ? e[0] : 0.0f;
                mb[u] = e[C];
                qb[u] = e[2 * C];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (nb[u] <= 0.0f) continue;
                const float nn = n + nb[u], d = mb[u] - mean;
                mean += d * (nb[u] / nn);
                m2 += qb[u] + d * d * (n * nb[u] / nn);
                n = nn;
            }
        }
        const float var = n > 0.0f ? m2 / n : 0.0f;
        const float sc = (p.gamma ? p.gamma[c] : 1.0f) / sqrtf(var + p.eps);
        p.affine[(size_t)view * 2 * C + c] = sc;
        p.affine[(size_t)view * 2 * C + C + c] = (p.beta ? p.beta[c] : 0.0f) - mean * sc;
    }
}

__global__ __launch_bounds__(256) void bn_views_apply_kernel(const float *x, const float *affine, const float *residual, float *out,
                                                              int rows_per_view, int C, long long total4, int relu)
{
    const int C4 = C >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        const long long row = e / C4;
        const int view = (int)(row / rows_per_view);
        const float4 v = reinterpret_cast<const float4 *>(x)[e];
        const float4 sc = *reinterpret_cast<const float4 *>(affine + (size_t)view * 2 * C + 4 * c4);
        const float4 sh = *reinterpret_cast<const float4 *>(affine + (size_t)view * 2 * C + C + 4 * c4);
        float4 y = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
        if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        if (residual) {
            const float4 rr = reinterpret_cast<const float4 *>(residual)[e];
            y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
        }
        reinterpret_cast<float4 *>(out)[e] = y;
    }
}

struct DwParams {
    const float *x, *w, *affine;     // w [K*K][C] tap-major; affine [V][2][C] or null
    float *out;
    int N, H, W, C, Ho, Wo, imgs_per_view, relu;
};

// PX output pixels of a row per thread: the K + (PX - 1) S input columns of a tap row are loaded (and BatchNorm'ed) once and
// shared by the PX outputs — 10 loads per output instead of 25 for the 5 x 5 stride-1 layers.  Every output still sums its taps in
// (dy, dx) order.
template <int K, int S, int PX>
__global__ __launch_bounds__(256) void dwconv_nhwc_kernel(DwParams p)
{
    constexpr int P = K / 2, COLS = K + (PX - 1) * S;
    const int C4 = p.C >> 2;
    const int wg = (p.Wo + PX - 1) / PX;
    const long long total = (long long)p.N * p.Ho * wg * C4;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c4 = (int)(e % C4);
    long long pix = e / C4;
    const int ox0 = (int)(pix % wg) * PX;
    pix /= wg;
    const int oy = (int)(pix % p.Ho);
    const int n = (int)(pix / p.Ho);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.affine) {
        const float *a = p.affine + (size_t)(n / p.imgs_per_view) * 2 * p.C + 4 * c4;
        sc = *reinterpret_cast<const float4 *>(a);
        sh = *reinterpret_cast<const float4 *>(a + p.C);
    }
    const float *img = p.x + (size_t)n * p.H * p.W * p.C + 4 * c4;
    const float *wq = p.w + 4 * c4;
    float4 acc[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < K; ++dy) {
        const int iy = oy * S + dy - P;
        if (iy < 0 || iy >= p.H) continue;
        float4 v[COLS];
#pragma unroll
        for (int cx = 0; cx < COLS; ++cx) {
            const int ix = ox0 * S + cx - P;
            v[cx] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ix >= 0 && ix < p.W) {
                float4 t = *reinterpret_cast<const float4 *>(img + ((size_t)iy * p.W + ix) * p.C);
                if (p.affine) {
                    t.x = t.x * sc.x + sh.x; t.y = t.y * sc.y + sh.y; t.z = t.z * sc.z + sh.z; t.w = t.w * sc.w + sh.w;
                    if (p.relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
                }
                v[cx] = t;
            }
        }
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            const float4 w = *reinterpret_cast<const float4 *>(wq + (size_t)(dy * K + dx) * p.C);
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const float4 t = v[j * S + dx];
                acc[j].x = fmaf(t.x, w.x, acc[j].x); acc[j].y = fmaf(t.y, w.y, acc[j].y);
                acc[j].z = fmaf(t.z, w.z, acc[j].z); acc[j].w = fmaf(t.w, w.w, acc[j].w);
            }
        }
    }
    float *orow = p.out + (((size_t)n * p.Ho + oy) * p.Wo) * p.C + 4 * c4;
#pragma unroll
    for (int j = 0; j < PX; ++j)
        if (ox0 + j < p.Wo) *reinterpret_cast<float4 *>(orow + (size_t)(ox0 + j) * p.C) = acc[j];
}

template <int K, int S, int PX>
void launch_dw(const DwParams &p, hipStream_t st)
{
    const long long total = (long long)p.N * p.Ho * ((p.Wo + PX - 1) / PX) * (p.C / 4);
    hipLaunchKernelGGL((dwconv_nhwc_kernel<K, S, PX>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, p);
}

bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" {

// A workgroup has 256 / (channels / 4) row lanes; a range is ~16 rows per lane (a thread's loop is a chain of load latencies,
// four in flight), at most 128 ranges per view (the last workgroup of a view merges them).
int eprecon_bn2d_views_chunks(int64_t rows_per_view, int channels)
{
    if (channels < 4) return 1;
    const int64_t lanes = 256 / (channels / 4) > 0 ? 256 / (channels / 4) : 1;
    const int64_t c = (rows_per_view + lanes * 16 - 1) / (lanes * 16);
    return (int)(c < 1 ? 1 : (c > 128 ? 128 : c));
}

size_t eprecon_bn2d_views_workspace_bytes(int views, int64_t rows_per_view, int channels)
{
    if (views <= 0 || channels <= 0) return 0;
    return (size_t)views * eprecon_bn2d_views_chunks(rows_per_view, channels) * 3 * channels * sizeof(float);
}

int eprecon_bn2d_views_stats_async(const float *x, int views, int64_t rows_per_view, int channels, const float *gamma,
                                   const float *beta, float eps, float *affine_out, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (views < 0 || rows_per_view < 0 || channels <= 0 || channels % 4 || channels > 480 || rows_per_view > 0x7fffffff) return EPRECON_ERR_ARG;
    if (views == 0) return EPRECON_OK;
    if (!x || !affine_out || !workspace || !aligned16(x) || !aligned16(affine_out)) return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_bn2d_views_workspace_bytes(views, rows_per_view, channels)) return EPRECON_ERR_WORKSPACE;
    BnViewsParams p;
    p.x = x; p.rows_per_view = (int)rows_per_view; p.C = channels; p.chunks = eprecon_bn2d_views_chunks(rows_per_view, channels);
    p.partial = (float *)workspace; p.gamma = gamma; p.beta = beta; p.eps = eps; p.affine = affine_out;
    hipLaunchKernelGGL(bn_views_stats_kernel, dim3((unsigned)p.chunks, (unsigned)views), dim3(256), 0, (hipStream_t)stream, p);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_views_finalize_kernel, dim3((unsigned)views, (unsigned)ceil_div(channels, 64)), dim3(64), 0, (hipStream_t)stream, p);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_bn2d_views_apply_async(const float *x, int views, int64_t rows_per_view, int channels, const float *affine, int relu,
                                   const float *residual, float *out, void *stream)
{
    if (views < 0 || rows_per_view < 0 || channels <= 0 || channels % 4 || rows_per_view > 0x7fffffff) return EPRECON_ERR_ARG;
    const long long total4 = (long long)views * rows_per_view * (channels / 4);
    if (total4 == 0) return EPRECON_OK;
    if (!x || !affine || !out || !aligned16(x) || !aligned16(out) || !aligned16(affine) || (residual && !aligned16(residual))) return EPRECON_ERR_ARG;
    const long long blocks = ceil_div(total4, 256 * 4);
    hipLaunchKernelGGL(bn_views_apply_kernel, dim3((unsigned)min(blocks, (long long)1 << 20)), dim3(256), 0, (hipStream_t)stream, x, affine,
                       residual, out, (int)rows_per_view, channels, total4, relu);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_dwconv2d_nhwc_async(const float *x, int n, int height, int width, int channels, const float *weight_taps, int ksize,
                                int stride, const float *affine, int imgs_per_view, int relu, float *out, void *stream)
{
    if (n < 0 || height <= 0 || width <= 0 || channels <= 0 || channels % 4 || (ksize != 3 && ksize != 5) || (stride != 1 && stride != 2) ||
        imgs_per_view <= 0)
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    if (!x || !weight_taps || !out || !aligned16(x) || !aligned16(out) || !aligned16(weight_taps) || (affine && !aligned16(affine)))
        return EPRECON_ERR_ARG;
    DwParams p;
    p.x = x; p.w = weight_taps; p.affine = affine; p.out = out; p.N = n; p.H = height; p.W = width; p.C = channels;
    p.Ho = (height + 2 * (ksize / 2) - ksize) / stride + 1;
    p.Wo = (width + 2 * (ksize / 2) - ksize) / stride + 1;
    p.imgs_per_view = imgs_per_view; p.relu = relu;
    const long long total = (long long)n * p.Ho * p.Wo * (channels / 4);
    if (total > 0x7fffffffll * 256) return EPRECON_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // (a padded input value enters an output's sum as fmaf(0, w, acc) = acc: the (dy, dx) order of a sum does not depend on PX)
    if (ksize == 3 && stride == 1) launch_dw<3, 1, 4>(p, st);
    else if (ksize == 3) launch_dw<3, 2, 2>(p, st);
    else if (stride == 1) launch_dw<5, 1, 4>(p, st);
    else launch_dw<5, 2, 2>(p, st);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"
