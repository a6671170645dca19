// Backward kernels of the sparse operators (the native path can also train).
//
//   spconv_wgrad_kernel    dW[k] = sum_i x[nbr[k][i]]^T dy[i]: the contraction runs over the voxel rows, so both
//                          operands are staged ROW-major in LDS exactly as they lie in HBM — for
//                          v_mfma_f32_32x32x2_f32 the A operand (Cin x rows) and the B operand (rows x Cout) are then
//                          both read with lane & 31 running along the channel axis (conflict-free, no transposes).
//                          The (-1) entries of a kernel map are compacted away per workgroup first (ballot ranks,
//                          deterministic order), so an offset that only 30 % of the voxels have costs 30 % of the MFMAs.
//                          Row chunks write partial products; wgrad_reduce_kernel sums them in chunk order: the
//                          result is deterministic (no float atomics).
//   invert_map_kernel      inv[k][j] = i for nbr[k][i] = j: the input gradient of a convolution is the same
//                          gather-GEMM over the inverted map with the transposed weights, i.e. the forward kernel.
//   devoxelize_bwd_kernel  d voxel_feat[idx8[p][c]] += w8[p][c] * d out[p]  (hardware float atomics)
//   gather_rows_scaled     d point_feat[p] = d voxel_mean[idx[p]] * scale[idx[p]]  (segment-mean backward)
#include "common.hpp"

namespace {

using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

constexpr int kRowBlock = 32;    // voxel rows per LDS stage (the contraction depth of one stage)
constexpr int kSubChunk = 2048;  // rows compacted at a time

struct WgradParams {
    const float *x; int ld_x;
    const float *dy; int ld_dy;
    const int32_t *nbr;  // [kvol][n_out] or nullptr (identity)
    int kvol; int64_t n_out;
    int cin, cout;
    int nchunks; int64_t chunk_rows;
    float *partial;  // [nchunks][kvol][cin][cout]
};

// The workgroup owns a (32 NI WI) x (32 NJ WJ) block of dW[k]: WI x WJ x WR = 4 waves, each with NI x NJ 32x32 MFMA tiles;
// the WR waves that share a tile split the rows of every stage (k-steps r = wr mod WR) and are summed through LDS in
// wave order at the end.  Narrow layers (the finest level runs 48 -> 24 channels over > 100k voxels) therefore keep all
// four waves on useful MFMAs instead of padding a 64 x 64 tile.  Global loads of the next row block are issued before
// the MFMA loop of the current one (register prefetch).
template <int NI, int NJ, int WI, int WJ, int WR>
__global__ __launch_bounds__(256) void spconv_wgrad_kernel(WgradParams p)
{
    static_assert(WI * WJ * WR == 4, "four waves");
    constexpr int TI = 32 * NI * WI, TJ = 32 * NJ * WJ;
    constexpr int SA = TI + 32, SG = TJ + 32;  // row pitches: the two half-waves read rows 32 banks apart
    constexpr int LA = kRowBlock * (TI / 4) / 256, LG = kRowBlock * (TJ / 4) / 256;  // float4 loads per thread and stage
    static_assert(LA >= 1 && LG >= 1, "tile too small for the staging loop");
    constexpr int RED = (WR > 1) ? (WR - 1) * WI * WJ * NI * NJ * 1024 : 1;
    constexpr int STAGE = kRowBlock * (SA + SG);
    __shared__ __align__(16) float sBuf[STAGE > RED ? STAGE : RED];
    float *sA = sBuf, *sG = sBuf + kRowBlock * SA;
    __shared__ int sIn[kSubChunk], sOut[kSubChunk];
    __shared__ int sWave[4];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r32 = lane & 31, half = lane >> 5;
    const int wr = wid % WR, wj = (wid / WR) % WJ, wi = wid / (WR * WJ);
    const int k = blockIdx.x % p.kvol, chunk = blockIdx.x / p.kvol;
    const int ci0 = blockIdx.y * TI, co0 = blockIdx.z * TJ;
    const int64_t row0 = (int64_t)chunk * p.chunk_rows;
    const int64_t row1 = row0 + p.chunk_rows < p.n_out ? row0 + p.chunk_rows : p.n_out;

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < NJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    auto load_row = [&](const float *base, int64_t row, int ld, int c0, int climit) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *src = base + row * ld + c0;
        const int left = climit - c0;
        if (left >= 4 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
            v = *reinterpret_cast<const float4 *>(src);
        } else {
            if (left > 0) v.x = src[0];
            if (left > 1) v.y = src[1];
            if (left > 2) v.z = src[2];
            if (left > 3) v.w = src[3];
        }
        return v;
    };

    for (int64_t sub0 = row0; sub0 < row1; sub0 += kSubChunk) {
        // ---- compaction of the valid (input row, output row) pairs of this sub-chunk, in row order ----
        int cnt = 0;
        const int sub_n = (int)(row1 - sub0 < kSubChunk ? row1 - sub0 : kSubChunk);
        for (int base = 0; base < sub_n; base += 256) {
            const int64_t i = sub0 + base + tid;
            int j = -1;
            if (base + tid < sub_n) j = p.nbr ? p.nbr[(int64_t)k * p.n_out + i] : (int)i;
            int tot;
            const int rank = ep::block_exclusive_rank<256>(j >= 0, sWave, tot);
            if (j >= 0) {
                sIn[cnt + rank] = j;
                sOut[cnt + rank] = (int)i;
            }
            cnt += tot;
            __syncthreads();
        }
        // ---- contraction over the compacted rows ----
        float4 ra[LA], rg[LG];
        auto fetch = [&](int rb) {
#pragma unroll
            for (int u = 0; u < LA; ++u) {
                const int e = tid + u * 256;
                const int r = e / (TI / 4), c4 = (e - r * (TI / 4)) * 4;
                ra[u] = rb + r < cnt ? load_row(p.x, sIn[rb + r], p.ld_x, ci0 + c4, p.cin) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < LG; ++u) {
                const int e = tid + u * 256;
                const int r = e / (TJ / 4), c4 = (e - r * (TJ / 4)) * 4;
                rg[u] = rb + r < cnt ? load_row(p.dy, sOut[rb + r], p.ld_dy, co0 + c4, p.cout) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        if (cnt > 0) fetch(0);
        for (int rb = 0; rb < cnt; rb += kRowBlock) {
#pragma unroll
            for (int u = 0; u < LA; ++u) {
                const int e = tid + u * 256;
                const int r = e / (TI / 4), c4 = (e - r * (TI / 4)) * 4;
                *reinterpret_cast<float4 *>(&sA[r * SA + c4]) = ra[u];
            }
#pragma unroll
            for (int u = 0; u < LG; ++u) {
                const int e = tid + u * 256;
                const int r = e / (TJ / 4), c4 = (e - r * (TJ / 4)) * 4;
                *reinterpret_cast<float4 *>(&sG[r * SG + c4]) = rg[u];
            }
            __syncthreads();
            if (rb + kRowBlock < cnt) fetch(rb + kRowBlock);
#pragma unroll 4
            for (int kk = wr; kk < kRowBlock / 2; kk += WR) {
                const int r = 2 * kk + half;
                float a[NI], b[NJ];
#pragma unroll
                for (int t = 0; t < NI; ++t) a[t] = sA[r * SA + (wi * NI + t) * 32 + r32];
#pragma unroll
                for (int t = 0; t < NJ; ++t) b[t] = sG[r * SG + (wj * NJ + t) * 32 + r32];
#pragma unroll
                for (int ti = 0; ti < NI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NJ; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    if (WR > 1) {  // the row groups of a tile are summed in wave order: wr = 1, 2, .. into wr = 0
        float *slot = sBuf + ((size_t)((wr - 1) * WI * WJ + wi * WJ + wj) * NI * NJ) * 1024;
        if (wr > 0) {
#pragma unroll
            for (int ti = 0; ti < NI; ++ti)
#pragma unroll
                for (int tj = 0; tj < NJ; ++tj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slot[((ti * NJ + tj) * 16 + r) * 64 + lane] = acc[ti][tj][r];
        }
        __syncthreads();
        if (wr > 0) return;
        for (int g = 1; g < WR; ++g) {
            const float *src = sBuf + ((size_t)((g - 1) * WI * WJ + wi * WJ + wj) * NI * NJ) * 1024;
#pragma unroll
            for (int ti = 0; ti < NI; ++ti)
#pragma unroll
                for (int tj = 0; tj < NJ; ++tj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ti][tj][r] += src[((ti * NJ + tj) * 16 + r) * 64 + lane];
        }
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *dst = p.partial + ((int64_t)chunk * p.kvol + k) * p.cin * p.cout;
#pragma unroll
    for (int ti = 0; ti < NI; ++ti)
#pragma unroll
        for (int tj = 0; tj < NJ; ++tj) {
            const int co = co0 + (wj * NJ + tj) * 32 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (wi * NI + ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ci < p.cin && co < p.cout) dst[(int64_t)ci * p.cout + co] = acc[ti][tj][r];
            }
        }
}

// dW[e] = sum over chunks, in a fixed order: 32 consecutive elements per workgroup, eight chunk lanes (lane g sums chunks
// g, g + 8, ...), then the lanes in ascending order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *partial, int nchunks, int64_t elems, float *dw)
{
    __shared__ float sPart[8][32];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + el;
    float s = 0.0f;
    if (e < elems)
        for (int c = g; c < nchunks; c += 8) s += partial[(int64_t)c * elems + e];
    sPart[g][el] = s;
    __syncthreads();
    if (g == 0 && e < elems) {
        float t = sPart[0][el];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += sPart[q][el];
        dw[e] = t;
    }
}

__global__ __launch_bounds__(256) void invert_map_kernel(const int32_t *nbr, int kvol, int64_t n_out, int64_t n_in,
                                                         int32_t *inv)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)kvol * n_out) return;
    const int k = (int)(e / n_out);
    const int j = nbr[e];
    if (j >= 0 && j < n_in) inv[(int64_t)k * n_in + j] = (int)(e - (int64_t)k * n_out);
}

__global__ __launch_bounds__(256) void devoxelize_bwd_kernel(const float *dout, int ld_o, const int32_t *idx8,
                                                             const float *w8, int64_t n, int channels, float *dfeat,
                                                             int ld_f)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * channels) return;
    const int64_t pt = e / channels;
    const int c = (int)(e - pt * channels);
    const float g = dout[pt * ld_o + c];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int v = idx8[pt * 8 + q];
        if (v >= 0) unsafeAtomicAdd(dfeat + (int64_t)v * ld_f + c, w8[pt * 8 + q] * g);
    }
}

// The same without atomics: the (point, corner) entries that reach a voxel come as a CSR list (eprecon_segment_lists_async over
// the flattened idx8: ascending entry index inside a voxel), one thread per (voxel, channel) adds its entries up in list order.
// Deterministic; the list depends on the corner table only and is shared by every layer that devoxelises with it.
__global__ __launch_bounds__(256) void devoxelize_bwd_csr_kernel(const float *dout, int ld_o, const float *w8, const int32_t *offsets,
                                                                 const int32_t *order, int64_t m, int channels, float *dfeat, int ld_f)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * channels) return;
    const int64_t v = e / channels;
    const int c = (int)(e - v * channels);
    float acc = 0.0f;
    for (int k = offsets[v]; k < offsets[v + 1]; ++k) {
        const int ent = order[k];
        acc = fmaf(w8[ent], dout[(int64_t)(ent >> 3) * ld_o + c], acc);
    }
    dfeat[v * ld_f + c] = acc;
}

__global__ __launch_bounds__(256) void gather_rows_scaled_kernel(const float *src, int ld_s, const int32_t *idx,
                                                                 const float *scale, int64_t n, int channels, float *dst,
                                                                 int ld_d)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * channels) return;
    const int64_t pt = e / channels;
    const int c = (int)(e - pt * channels);
    const int v = idx[pt];
    dst[pt * ld_d + c] = v >= 0 ? src[(int64_t)v * ld_s + c] * (scale ? scale[v] : 1.0f) : 0.0f;
}

// tile of dW[k] one workgroup owns, by layer width
void wgrad_tile(int cin, int cout, int &ti, int &tj)
{
    tj = cout <= 32 ? 32 : (cout <= 64 ? 64 : 128);
    ti = cin <= 32 ? 32 : ((cin <= 64 || tj < 128) ? 64 : 128);
}

int wgrad_chunks(int kvol, int64_t n_out, int cin, int cout)
{
    int ti, tj;
    wgrad_tile(cin, cout, ti, tj);
    const int64_t tiles = ep::ceil_div(cin, ti) * ep::ceil_div(cout, tj);
    int64_t want = ep::ceil_div(768, (int64_t)kvol * tiles);
    const int64_t most = ep::ceil_div(n_out, 256);
    if (want > most) want = most;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace

extern "C" {

size_t eprecon_sparse_conv_wgrad_workspace_bytes(int kvol, int64_t n_out, int cin, int cout)
{
    const int nch = wgrad_chunks(kvol, n_out, cin, cout);
    return (size_t)nch * kvol * cin * cout * sizeof(float);
}

int eprecon_sparse_conv_wgrad_async(const float *x, int ld_x, const float *dy, int ld_dy, const int32_t *nbr, int kvol,
                                    int64_t n_out, int cin, int cout, float *dweight, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    if (!x || !dy || !dweight || kvol < 1 || cin < 1 || cout < 1 || n_out < 0 || (!nbr && kvol != 1)) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t elems = (int64_t)kvol * cin * cout;
    if (n_out == 0) {
        EP_HIP_CHECK(hipMemsetAsync(dweight, 0, elems * sizeof(float), st));
        return EPRECON_OK;
    }
    if (workspace_bytes < eprecon_sparse_conv_wgrad_workspace_bytes(kvol, n_out, cin, cout) || !workspace)
        return EPRECON_ERR_WORKSPACE;
    int ti, tj;
    wgrad_tile(cin, cout, ti, tj);
    WgradParams p;
    p.x = x; p.ld_x = ld_x; p.dy = dy; p.ld_dy = ld_dy; p.nbr = nbr; p.kvol = kvol; p.n_out = n_out;
    p.cin = cin; p.cout = cout;
    p.nchunks = wgrad_chunks(kvol, n_out, cin, cout);
    p.chunk_rows = ep::ceil_div(ep::ceil_div(n_out, p.nchunks), 256) * 256;
    p.nchunks = (int)ep::ceil_div(n_out, p.chunk_rows);
    p.partial = (float *)workspace;
    dim3 grid((unsigned)(p.nchunks * kvol), (unsigned)ep::ceil_div(cin, ti), (unsigned)ep::ceil_div(cout, tj));
#define EP_WGRAD(NI, NJ, WI, WJ, WR) hipLaunchKernelGGL((spconv_wgrad_kernel<NI, NJ, WI, WJ, WR>), grid, dim3(256), 0, st, p)
    if (ti == 32 && tj == 32) EP_WGRAD(1, 1, 1, 1, 4);
    else if (ti == 64 && tj == 32) EP_WGRAD(1, 1, 2, 1, 2);
    else if (ti == 32 && tj == 64) EP_WGRAD(1, 1, 1, 2, 2);
    else if (ti == 64 && tj == 64) EP_WGRAD(1, 1, 2, 2, 1);
    else if (ti == 32 && tj == 128) EP_WGRAD(1, 2, 1, 2, 2);
    else if (ti == 64 && tj == 128) EP_WGRAD(1, 2, 2, 2, 1);
    else EP_WGRAD(2, 2, 2, 2, 1);
#undef EP_WGRAD
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ep::ceil_div(elems, 32)), dim3(256), 0, st, p.partial,
                       p.nchunks, elems, dweight);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_invert_map_async(const int32_t *nbr, int kvol, int64_t n_out, int64_t n_in, int32_t *inv, void *stream)
{
    if (kvol < 1 || n_out < 0 || n_in < 0 || (n_in > 0 && !inv) || (n_out > 0 && !nbr)) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n_in > 0) EP_HIP_CHECK(hipMemsetAsync(inv, 0xFF, (size_t)kvol * n_in * sizeof(int32_t), st));
    if (n_out > 0 && n_in > 0) {
        hipLaunchKernelGGL(invert_map_kernel, dim3((unsigned)ep::ceil_div((int64_t)kvol * n_out, 256)), dim3(256), 0, st,
                           nbr, kvol, n_out, n_in, inv);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}

int eprecon_devoxelize_backward_async(const float *dout, int ld_out, const int32_t *idx8, const float *weight8, int64_t n,
                                      int channels, int64_t n_voxels, float *dvoxel_feat, int ld_feat, void *stream)
{
    if (n < 0 || channels < 0 || n_voxels < 0 || (n_voxels > 0 && channels > 0 && !dvoxel_feat)) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n_voxels > 0 && channels > 0)
        EP_HIP_CHECK(hipMemset2DAsync(dvoxel_feat, (size_t)ld_feat * sizeof(float), 0, (size_t)channels * sizeof(float),
                                      (size_t)n_voxels, st));
    if (n > 0 && channels > 0 && n_voxels > 0) {
        if (!dout || !idx8 || !weight8) return EPRECON_ERR_ARG;
        hipLaunchKernelGGL(devoxelize_bwd_kernel, dim3((unsigned)ep::ceil_div(n * channels, 256)), dim3(256), 0, st, dout,
                           ld_out, idx8, weight8, n, channels, dvoxel_feat, ld_feat);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}

int eprecon_devoxelize_backward_csr_async(const float *dout, int ld_out, const float *weight8, const int32_t *offsets,
                                          const int32_t *order, int64_t n_voxels, int channels, float *dvoxel_feat, int ld_feat,
                                          void *stream)
{
    if (channels < 0 || n_voxels < 0) return EPRECON_ERR_ARG;
    if (n_voxels == 0 || channels == 0) return EPRECON_OK;
    if (!dout || !weight8 || !offsets || !order || !dvoxel_feat || n_voxels * channels > 0x7fffffffll * 256) return EPRECON_ERR_ARG;
    hipLaunchKernelGGL(devoxelize_bwd_csr_kernel, dim3((unsigned)ep::ceil_div(n_voxels * channels, 256)), dim3(256), 0,
                       (hipStream_t)stream, dout, ld_out, weight8, offsets, order, n_voxels, channels, dvoxel_feat, ld_feat);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_gather_rows_scaled_async(const float *src, int ld_src, const int32_t *idx, const float *scale, int64_t n,
                                     int channels, float *dst, int ld_dst, void *stream)
{
    if (n < 0 || channels < 0) return EPRECON_ERR_ARG;
    if (n == 0 || channels == 0) return EPRECON_OK;
    if (!src || !idx || !dst) return EPRECON_ERR_ARG;
    hipLaunchKernelGGL(gather_rows_scaled_kernel, dim3((unsigned)ep::ceil_div(n * channels, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, ld_src, idx, scale, n, channels, dst, ld_dst);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"
