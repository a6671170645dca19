// Small dense-grid helpers of the coarse-to-fine loop for gfx950.
//
//   init_select   sigmoid(logit) > thr on the valid 48^3 voxels -> OR-pool 2^3 -> erode -> dilate x2
//                 -> raster-order coordinates * 4           models/neucon_network.py:264,298-318
//                 (marking is one thread per voxel; the 24^3 morphology runs on bit columns, one workgroup)
//   upsample      every voxel -> its 8 children, parent-major, features replicated
//                                                            models/neucon_network.py:193-214
#include "common.hpp"

namespace {
using namespace ep;

constexpr int kSelThreads = 1024;

// one thread per valid voxel: mark the coarse cell of every voxel with sigmoid(logit) > thr
// (byte stores of the same value: race-free)
__global__ __launch_bounds__(256) void init_mark_kernel(const float *logit, const int4 *coords, int n, float thr,
                                                        int batch, int D, int cell, unsigned char *marks)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = coords[i];
    if (c.x < 0 || c.x >= batch) return;
    const float sig = 1.0f / (1.0f + expf(-logit[i]));
    if (sig > thr) {
        const int x = c.y / cell, y = c.z / cell, z = c.w / cell;
        if (x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D) marks[(size_t)c.x * D * D * D + (x * D + y) * D + z] = 1;
    }
}

// Morphology on bit columns: the D (<= 32) cells of one (x, y) column are one 32-bit word, bit z.
// A zero-padded 3^3 box erosion is AND over the 9 neighbour columns of (w & w<<1 & w>>1); the
// dilation is the same with OR.  One thread per column; the whole volume is D*D words in LDS.
__device__ __forceinline__ uint32_t box_columns(const uint32_t *vol, int D, int x, int y, bool erode, uint32_t zmask)
{
    uint32_t r = erode ? zmask : 0u;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy) {
            const int a = x + dx, b = y + dy;
            const uint32_t w = (a >= 0 && a < D && b >= 0 && b < D) ? vol[a * D + b] : 0u;
            if (erode)
                r &= w & (w << 1) & (w >> 1);
            else
                r |= w | (w << 1) | (w >> 1);
        }
    return r & zmask;
}

__global__ __launch_bounds__(kSelThreads) void init_select_kernel(const unsigned char *marks, int batch, int D,
                                                                  int out_scale, int4 *out_coords,
                                                                  int32_t *n_out_dev)
{
    __shared__ uint32_t va[1024], vb[1024];
    __shared__ int sWave[kSelThreads / kWave];
    const int cols = D * D, cells = cols * D;
    const uint32_t zmask = D >= 32 ? 0xffffffffu : ((1u << D) - 1u);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    const int x = tid / D, y = tid - x * D;
    const bool live = tid < cols;
    int written = 0;
    for (int b = 0; b < batch; ++b) {
        uint32_t w = 0;
        if (live)
            for (int z = 0; z < D; ++z) w |= marks[(size_t)b * cells + tid * D + z] ? (1u << z) : 0u;
        if (live) va[tid] = w;
        __syncthreads();
        if (live) vb[tid] = box_columns(va, D, x, y, true, zmask);   // erode
        __syncthreads();
        if (live) va[tid] = box_columns(vb, D, x, y, false, zmask);  // dilate
        __syncthreads();
        const uint32_t fin = live ? box_columns(va, D, x, y, false, zmask) : 0u;  // dilate
        // raster-order compaction: columns in (x, y) order, bits in ascending z
        const int mine = __popc(fin);
        int s = mine;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int t = __shfl_up(s, d);
            if (lane >= d) s += t;
        }
        if (lane == kWave - 1) sWave[wid] = s;
        __syncthreads();
        int off = written + s - mine, tot = 0;
        for (int k = 0; k < kSelThreads / kWave; ++k) {
            const int c = sWave[k];
            if (k < wid) off += c;
            tot += c;
        }
        uint32_t bits = fin;
        while (bits) {
            const int z = __ffs(bits) - 1;
            bits &= bits - 1;
            out_coords[off++] = make_int4(b, x * out_scale, y * out_scale, z * out_scale);
        }
        if (tid == 0) n_out_dev[1 + b] = tot;
        written += tot;
        __syncthreads();
    }
    if (tid == 0) n_out_dev[0] = written;
}

__global__ __launch_bounds__(256) void upsample_coords_kernel(const int4 *coords, int n, int interval,
                                                              int4 *up)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * 8) return;
    const int i = e >> 3, k = e & 7;
    // child order (models/neucon_network.py:204-209): 0, x, y, z, xy, xz, yz, xyz
    const int dx = (0xB2 >> k) & 1;  // k in {1,4,5,7}
    const int dy = (0xD4 >> k) & 1;  // k in {2,4,6,7}
    const int dz = (0xE8 >> k) & 1;  // k in {3,5,6,7}
    int4 c = coords[i];
    c.y += dx * interval;
    c.z += dy * interval;
    c.w += dz * interval;
    up[e] = c;
}

__global__ __launch_bounds__(256) void upsample_feat_kernel(const float *feat, int n, int C, int ld,
                                                            float *up)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n * 8 * C;
    if (e >= total) return;
    const int c = (int)(e % C);
    const size_t row = e / C;
    up[e] = feat[(row >> 3) * ld + c];
}

}  // namespace

extern "C" {

size_t eprecon_init_select_workspace_bytes(int batch, int dim)
{
    return align_up((size_t)batch * dim * dim * dim, 256);
}

int eprecon_init_select_async(const float *logit, const int32_t *coords, int64_t n, float threshold,
                              int batch, int dim, int cell, int32_t *out_coords, int32_t *n_out_dev,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || batch <= 0 || dim <= 0 || dim > 32 || cell <= 0 || !out_coords || !n_out_dev || !workspace ||
        (n > 0 && (!logit || !coords)))
        return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_init_select_workspace_bytes(batch, dim)) return EPRECON_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int cells = dim * dim * dim;
    unsigned char *marks = reinterpret_cast<unsigned char *>(workspace);
    EP_HIP_CHECK(hipMemsetAsync(marks, 0, (size_t)batch * cells, st));
    if (n > 0) {
        hipLaunchKernelGGL(init_mark_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, logit,
                           reinterpret_cast<const int4 *>(coords), (int)n, threshold, batch, dim, cell, marks);
        EP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(init_select_kernel, dim3(1), dim3(kSelThreads), 0, st, (const unsigned char *)marks, batch,
                       dim, cell, reinterpret_cast<int4 *>(out_coords), n_out_dev);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_upsample_async(const float *feat, int ld_feat, const int32_t *coords, int64_t n, int channels,
                           int interval, float *up_feat, int32_t *up_coords, void *stream)
{
    // (up_coords null with channels > 0: the children were written by an earlier call, eprecon_spvcnn_points_dn_async)
    if (n < 0 || channels < 0 || interval <= 0 || (n > 0 && (!coords || (!up_coords && channels == 0)))) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipStream_t st = (hipStream_t)stream;
    if (up_coords) {
        hipLaunchKernelGGL(upsample_coords_kernel, dim3((unsigned)ceil_div(n * 8, 256)), dim3(256), 0, st,
                           reinterpret_cast<const int4 *>(coords), (int)n, interval,
                           reinterpret_cast<int4 *>(up_coords));
        EP_LAUNCH_CHECK();
    }
    if (channels > 0) {
        if (!feat || !up_feat || ld_feat < channels) return EPRECON_ERR_ARG;
        const size_t total = (size_t)n * 8 * channels;
        hipLaunchKernelGGL(upsample_feat_kernel, dim3((unsigned)ceil_div((int64_t)total, 256)), dim3(256), 0,
                           st, feat, (int)n, channels, ld_feat, up_feat);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// x2 bilinear upsampling of channels-last maps (F.interpolate(scale_factor=2, mode="bilinear",
// align_corners=False) in Occupancy_Initialization.feat_fusion_pre,
// models/occupancy_initialization.py:46): in f32[n, h, w, c] -> out f32[n, 2h, 2w, c].
// src = max(0, (dst + 0.5) / 2 - 0.5), taps clamped to the last row / column.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void upsample2x_nhwc_kernel(const float4 *in, float4 *out, int n, int h, int w,
                                                              int c4)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n * 4 * h * w * c4;
    if (e >= total) return;
    const int c = (int)(e % c4);
    size_t r = e / c4;
    const int ox = (int)(r % (2 * w));
    r /= 2 * w;
    const int oy = (int)(r % (2 * h));
    const int img = (int)(r / (2 * h));
    const float sx = fmaxf(((float)ox + 0.5f) * 0.5f - 0.5f, 0.0f);
    const float sy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.0f);
    const int x0 = (int)sx, y0 = (int)sy;
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const float lx = sx - (float)x0, ly = sy - (float)y0;
    const float4 *base = in + (size_t)img * h * w * c4;
    const float4 a = base[((size_t)y0 * w + x0) * c4 + c], b = base[((size_t)y0 * w + x1) * c4 + c];
    const float4 d = base[((size_t)y1 * w + x0) * c4 + c], f = base[((size_t)y1 * w + x1) * c4 + c];
    const float w00 = (1.0f - ly) * (1.0f - lx), w01 = (1.0f - ly) * lx, w10 = ly * (1.0f - lx), w11 = ly * lx;
    float4 o;
    o.x = w00 * a.x + w01 * b.x + w10 * d.x + w11 * f.x;
    o.y = w00 * a.y + w01 * b.y + w10 * d.y + w11 * f.y;
    o.z = w00 * a.z + w01 * b.z + w10 * d.z + w11 * f.z;
    o.w = w00 * a.w + w01 * b.w + w10 * d.w + w11 * f.w;
    out[e] = o;
}
}  // namespace

extern "C" int eprecon_upsample2x_nhwc_async(const float *in, float *out, int n, int h, int w, int channels,
                                             void *stream)
{
    if (!in || !out || n <= 0 || h <= 0 || w <= 0 || channels <= 0 || channels % 4) return EPRECON_ERR_ARG;
    const size_t total = (size_t)n * 4 * h * w * (channels / 4);
    hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3((unsigned)ceil_div((int64_t)total, 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), n, h,
                       w, channels / 4);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

// ---------------------------------------------------------------------------------------------
// Sparsify for the next stage (models/neucon_network.py:454-507) as one call: occupancy = occ > thr; per batch element the
// occupied voxels and the occupied voxels whose target is occupied (the guards' counts); the kept rows of coords / tsdf /
// occ / feat_all compacted in row order (what torch.nonzero + four index_select + a cat produce) — three launches and ONE
// host read instead of ~15 launches and two.  The random sub-sampling branch (:477-484) edits the occupancy on the host side
// and then takes the reference's own sequence of torch calls (eprecon_amd/neucon_network.py).
//   counts int32[1 + 2 * batch]: [0] kept rows, [1 + b] occupied in batch b, [1 + batch + b] occupied & target in batch b
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int kSpTile = 256;
constexpr int kSpMaxBatch = 32;

__global__ __launch_bounds__(kSpTile) void sparsify_count_kernel(const float *occ, int ld_occ, float thr, const unsigned char *tgt,
                                                                const int4 *coords, int n, int batch, int *tile_tot, int *counts)
{
    const int row = blockIdx.x * kSpTile + threadIdx.x;
    const bool keep = row < n && occ[(size_t)row * ld_occ] > thr;
    const unsigned long long m = __ballot(keep);
    __shared__ int sTot[kSpTile / 64];
    __shared__ int sCnt[2 * kSpMaxBatch];
    if (threadIdx.x < 2 * kSpMaxBatch) sCnt[threadIdx.x] = 0;
    if ((threadIdx.x & 63) == 0) sTot[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    // per-batch counts: ballots per wave -> LDS -> ONE global atomic per workgroup and batch element (a same-address atomic
    // per kept row serialises: 180k of them cost ~0.5 ms; integer sums do not depend on the order)
    const int b = keep ? coords[row].x : -1;
    const bool t = keep && (!tgt || tgt[row]);
    for (int bb = 0; bb < batch; ++bb) {
        const unsigned long long mb = __ballot(b == bb), mt = __ballot(b == bb && t);
        if ((threadIdx.x & 63) == 0 && mb) {
            atomicAdd(&sCnt[bb], __popcll(mb));
            if (mt) atomicAdd(&sCnt[kSpMaxBatch + bb], __popcll(mt));
        }
    }
    __syncthreads();
    if (threadIdx.x < batch) {
        if (sCnt[threadIdx.x]) atomicAdd(counts + 1 + threadIdx.x, sCnt[threadIdx.x]);
        if (sCnt[kSpMaxBatch + threadIdx.x]) atomicAdd(counts + 1 + batch + threadIdx.x, sCnt[kSpMaxBatch + threadIdx.x]);
    }
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = sTot[0] + sTot[1] + sTot[2] + sTot[3];
}

// exclusive scan of the tile totals in place (one workgroup; <= a few thousand tiles), total -> counts[0]
__global__ __launch_bounds__(1024) void sparsify_scan_kernel(int *tile_tot, int ntiles, int *counts)
{
    __shared__ int sWave[16];
    __shared__ int sCarry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sCarry = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += 1024) {
        const int i = base + tid;
        const int v = i < ntiles ? tile_tot[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += sWave[w];
        const int carry = sCarry;
        if (i < ntiles) tile_tot[i] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 1023) sCarry = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) counts[0] = sCarry;
}

// kept row r -> output row tile_off[tile] + (kept rows before it in the tile); one thread per row writes the small columns,
// the wide feature row is copied by the row's wave cooperatively
__global__ __launch_bounds__(kSpTile) void sparsify_compact_kernel(const float *occ, int ld_occ, float thr, const int4 *coords,
                                                                  const float *tsdf, int ld_tsdf, const float *feat, int ld_feat,
                                                                  int c_all, int c_feat, int n, const int *tile_off, int4 *out_coords,
                                                                  float *out_tsdf, float *out_occ, float *out_all, float *out_feat)
{
    const int row = blockIdx.x * kSpTile + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float o = row < n ? occ[(size_t)row * ld_occ] : 0.0f;
    const bool keep = row < n && o > thr;
    const unsigned long long m = __ballot(keep);
    __shared__ int sTot[kSpTile / 64];
    if (lane == 0) sTot[wave] = __popcll(m);
    __syncthreads();
    int base = tile_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += sTot[w];
    const int dst = base + __popcll(m & ((1ull << lane) - 1ull));
    const float t = keep ? tsdf[(size_t)row * ld_tsdf] : 0.0f;
    if (keep) {
        out_coords[dst] = coords[row];
        out_tsdf[dst] = t;
        out_occ[dst] = o;
        out_feat[(size_t)dst * (c_feat + 2) + c_feat] = t;
        out_feat[(size_t)dst * (c_feat + 2) + c_feat + 1] = o;
    }
    // feature rows: the wave walks its kept rows, 64 lanes over the channels
    unsigned long long rest = m;
    while (rest) {
        const int l = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        const int src = blockIdx.x * kSpTile + wave * 64 + l;
        const int d = __shfl(dst, l);
        for (int c = lane; c < c_all; c += 64) {
            const float v = feat[(size_t)src * ld_feat + c];
            out_all[(size_t)d * c_all + c] = v;
            if (c < c_feat) out_feat[(size_t)d * (c_feat + 2) + c] = v;
        }
    }
}
}  // namespace

extern "C" size_t eprecon_sparsify_workspace_bytes(int64_t n)
{
    return align_up((size_t)(ceil_div(n > 0 ? n : 1, (int64_t)kSpTile) + 1) * sizeof(int), 256);
}

extern "C" int eprecon_sparsify_async(const float *occ, int ld_occ, float threshold, const unsigned char *target,
                                      const int32_t *coords, const float *tsdf, int ld_tsdf, const float *feat_all, int ld_feat,
                                      int c_all, int c_feat, int64_t n, int batch, int32_t *out_coords, float *out_tsdf,
                                      float *out_occ, float *out_all, float *out_feat, int32_t *counts, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    if (n < 0 || n > 0x7fffffff || batch <= 0 || batch > kSpMaxBatch || c_all <= 0 || c_feat < 0 || c_feat > c_all || ld_occ < 1 || ld_tsdf < 1 ||
        ld_feat < c_all || !counts || !workspace || (n > 0 && (!occ || !coords || !tsdf || !feat_all || !out_coords || !out_tsdf ||
                                                               !out_occ || !out_all || !out_feat)))
        return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_sparsify_workspace_bytes(n)) return EPRECON_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    EP_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)(1 + 2 * batch) * sizeof(int32_t), st));
    if (n == 0) return EPRECON_OK;
    int *tile_tot = reinterpret_cast<int *>(workspace);
    const int ntiles = (int)ceil_div(n, (int64_t)kSpTile);
    hipLaunchKernelGGL(sparsify_count_kernel, dim3(ntiles), dim3(kSpTile), 0, st, occ, ld_occ, threshold, target,
                       reinterpret_cast<const int4 *>(coords), (int)n, batch, tile_tot, counts);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(sparsify_scan_kernel, dim3(1), dim3(1024), 0, st, tile_tot, ntiles, counts);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(sparsify_compact_kernel, dim3(ntiles), dim3(kSpTile), 0, st, occ, ld_occ, threshold,
                       reinterpret_cast<const int4 *>(coords), tsdf, ld_tsdf, feat_all, ld_feat, c_all, c_feat, (int)n,
                       (const int *)tile_tot, reinterpret_cast<int4 *>(out_coords), out_tsdf, out_occ, out_all, out_feat);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}
