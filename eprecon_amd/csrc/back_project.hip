// Multi-view back-projection for gfx950 (MI355X): image-feature pyramid -> voxel list.
//
// Replaces Back_Project.forward (models/occupancy_initialization.py:189-261 of the reference),
// ops/back_project.py:5-80 and the sample + mean/variance block of
// Occupancy_Initialization.forward (models/occupancy_initialization.py:79-128).
//
// Pipeline per call (all on the caller's stream):
//   [nchw_to_nhwc]   re-layout of the V*B feature maps to channels-last, so that the C channels of
//                    one bilinear tap are one contiguous 4*C-byte run (skipped when the caller
//                    already holds channels-last maps);
//   bp_count         one thread per voxel: 9 projections, visible-view count (float, all N),
//                    per-block valid totals by wave ballot + popcount, per-batch valid counts;
//   bp_scan          exclusive scan of the block totals (one workgroup) -> output offsets, n_valid;
//   bp_gather        phase 1, one thread per voxel: re-project, stable in-block compaction by
//                    ballot/prefix-sum, pixel coordinates of every (voxel, view) staged in LDS;
//                    phase 2, one thread per (valid voxel, 4-channel group): bilinear gather of
//                    the visible views with 16-byte loads, mean (or two-sweep variance) in
//                    registers, fully coalesced 16-byte stores in compacted order.
//   [bp_depth_norm]  ops.back_project's extra normalised-depth channel.
//
// Roofline: HBM/L2 bandwidth (about 2 flop per gathered byte).  Algorithmic bytes per call
//   16 N (coords in) + 4 N (count out) + 4 V C H W (maps, once) + n_valid (4 C + 16) (rows out).
// Arithmetic contract (bit-exact indices): see oracle/c/back_project_oracle.c and DESIGN.md.
#include <stdlib.h>

#include "common.hpp"

namespace {
using namespace ep;

// Variants measured on MI355X (tools/ab_backproject.py, dense 96^3, C = 24, 120x160; gather kernel / whole op).
// Only the first two are still in the build (the second is the default; the first takes C % 4 != 0); the others were
// parity-green experiments of rounds 1 / 2 and were removed again (git history: "back_project:" commits):
//   input-order tiles, 4-channel lanes, taps recomputed      134 us / 0.26 ms  (bp_gather_kernel)
//   + per-pair taps in LDS, cheap projection, buffer loads   122 us            (bp_gather_mlp_kernel, default)
//   + 2 / 3 / 4 views of loads in flight per lane            126 / 129 / 138 us (no latency to hide)
//   brick-sorted tiles, 4-channel lanes                      118 us / 0.27 ms  (binning costs what it saves)
//   8-channel lanes + per-pair taps in LDS + buffer loads    173-185 us        (strided 32-byte lanes)
//   128-byte padded pixel stride                             133 us            (neutral)
//   LDS image patches per view, barrier per view             206 us
//   LDS image patches, all views staged at once (64 voxels)  313 us
//   pixel-pair records [pix x | pix x+1] (192-byte aligned runs, 12 lanes per voxel, 6 instead of 8 L1
//   segments per (voxel, view); 2x the map footprint)         165 us
// PMC: 54.7 M vector-L1 accesses for 2.38 M wave loads = 23 per instruction: a 96-byte tap (24 channels)
// always touches two 64-byte L1 segments, and at one segment per clock per CU that is 89 of the 116 us: the
// L1 access rate bounds the kernel, not latency (more loads in flight do not help), not VALU (-43 % VALU
// bought 134 -> 122 us) and not HBM.

struct BpParams {
    const int32_t *coords;
    int n;
    const float *origin;
    int batch;
    float voxel_size;
    const float *feats_nhwc;  // [V*B][H*W][C]
    const float *krcam;       // [V*B][16]
    int V, C, H, W;
    int Cs;  // pixel stride of the channels-last maps in floats (>= C)
    int min_view;
    float *out_feats;
    float *out_mean;
    int32_t *out_coords;
    float *count;
    float *out_grid;
    uint8_t *out_mask;
    int32_t *n_valid_dev;  // [1 + B]
    int32_t *block_offsets;
    int xcd_slabs;  // 1 (default): every XCD walks one contiguous range of tiles (ep::xcd_remap); 0: tile = hardware block id
};

struct Proj {
    float gx, gy, pz;
    bool vis;
};

// P: rows 0..2 of a 4x4 row-major projection (12 floats).  k-ordered fma chain == torch CPU bmm
// == fp32 MFMA accumulation order (see the oracle header for the evidence).
__device__ __forceinline__ Proj project(const float *P, float X, float Y, float Z, float wm1,
                                        float hm1)
{
    const float px = __fmaf_rn(P[3], 1.0f, __fmaf_rn(P[2], Z, __fmaf_rn(P[1], Y, __fmul_rn(P[0], X))));
    const float py = __fmaf_rn(P[7], 1.0f, __fmaf_rn(P[6], Z, __fmaf_rn(P[5], Y, __fmul_rn(P[4], X))));
    const float pz = __fmaf_rn(P[11], 1.0f, __fmaf_rn(P[10], Z, __fmaf_rn(P[9], Y, __fmul_rn(P[8], X))));
    const float u = __fdiv_rn(px, pz);
    const float v = __fdiv_rn(py, pz);
    Proj r;
    r.gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, u), wm1), 1.0f);
    r.gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, v), hm1), 1.0f);
    r.pz = pz;
    r.vis = (fabsf(r.gx) <= 1.0f) && (fabsf(r.gy) <= 1.0f) && (pz > 0.0f);
    return r;
}

// Cheap projection with bit-exact visibility.  The fma chains for (px, py, pz) are the contract's;
// the three IEEE divisions per axis are replaced by one v_rcp_f32 + Newton step, and the exact
// sequence is re-evaluated only when the cheap normalised coordinate lands within 1e-4 of the
// frustum boundary |g| = 1 (the cheap value is within 1e-6 of the exact one, so outside that band
// both agree).  u, v are the pixel coordinates used for sampling (<= 1e-5 px from the reference's
// grid -> pixel round trip), clamped into the image.
struct ProjFast {
    float u, v, pz;
    bool vis;
};

__device__ __forceinline__ ProjFast project_fast(const float *P, float X, float Y, float Z, float wm1, float hm1,
                                                 float kx, float ky)
{
    const float px = __fmaf_rn(P[3], 1.0f, __fmaf_rn(P[2], Z, __fmaf_rn(P[1], Y, __fmul_rn(P[0], X))));
    const float py = __fmaf_rn(P[7], 1.0f, __fmaf_rn(P[6], Z, __fmaf_rn(P[5], Y, __fmul_rn(P[4], X))));
    const float pz = __fmaf_rn(P[11], 1.0f, __fmaf_rn(P[10], Z, __fmaf_rn(P[9], Y, __fmul_rn(P[8], X))));
    float r = __builtin_amdgcn_rcpf(pz);
    r = r * fmaf(-pz, r, 2.0f);
    const float u = px * r, v = py * r;
    const float gx = fmaf(u, kx, -1.0f), gy = fmaf(v, ky, -1.0f);
    ProjFast o;
    o.pz = pz;
    const bool near_edge = fabsf(fabsf(gx) - 1.0f) < 1e-4f || fabsf(fabsf(gy) - 1.0f) < 1e-4f;
    if (near_edge) {
        const float ue = __fdiv_rn(px, pz), ve = __fdiv_rn(py, pz);
        const float gxe = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ue), wm1), 1.0f);
        const float gye = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ve), hm1), 1.0f);
        o.vis = (fabsf(gxe) <= 1.0f) && (fabsf(gye) <= 1.0f) && (pz > 0.0f);
    } else {
        o.vis = (fabsf(gx) <= 1.0f) && (fabsf(gy) <= 1.0f) && (pz > 0.0f);
    }
    o.u = fminf(fmaxf(u, 0.0f), wm1);
    o.v = fminf(fmaxf(v, 0.0f), hm1);
    return o;
}

__device__ __forceinline__ void voxel_centre(const int4 c, const float *origin, float vs, float &X,
                                             float &Y, float &Z)
{
    // float(c) * voxel_size + origin: separate multiply and add (models/occupancy_initialization.py:213)
    X = __fadd_rn(__fmul_rn((float)c.y, vs), origin[3 * c.x + 0]);
    Y = __fadd_rn(__fmul_rn((float)c.z, vs), origin[3 * c.x + 1]);
    Z = __fadd_rn(__fmul_rn((float)c.w, vs), origin[3 * c.x + 2]);
}

__device__ __forceinline__ void stage_matrices(float *sP, const float *krcam, int nmat, int tid,
                                               int nthreads)
{
    for (int i = tid; i < nmat * 12; i += nthreads) {
        const int m = i / 12, e = i - m * 12;
        sP[i] = krcam[m * 16 + e];
    }
}

// ---------------------------------------------------------------------------------------------
// K1: visible-view count for every voxel, valid totals per block and per batch element
// ---------------------------------------------------------------------------------------------
// One thread per voxel, 256 per workgroup.  The valid totals are produced per TILE of VOX consecutive
// voxels (VOX = 256, 64 or 16: the tile the gather kernel hands to one workgroup).
template <int VOX>
__global__ __launch_bounds__(256) void bp_count_kernel(BpParams p, int32_t *tile_sums, int32_t *blk_batch)
{
    constexpr int BLOCK = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sP = reinterpret_cast<float *>(smem);
    int *sBatch = reinterpret_cast<int *>(sP + p.V * p.batch * 12);
    int *sWave = sBatch + p.batch;
    const int tid = threadIdx.x;
    stage_matrices(sP, p.krcam, p.V * p.batch, tid, BLOCK);
    for (int b = tid; b < p.batch; b += BLOCK) sBatch[b] = 0;
    __syncthreads();

    const int i = blockIdx.x * BLOCK + tid;
    bool valid = false;
    int vbatch = 0;
    if (i < p.n) {
        const int4 c = reinterpret_cast<const int4 *>(p.coords)[i];
        int cnt = 0;
        const bool in_range = c.x >= 0 && c.x < p.batch;
        if (in_range) {
            float X, Y, Z;
            voxel_centre(c, p.origin, p.voxel_size, X, Y, Z);
            const float wm1 = (float)(p.W - 1), hm1 = (float)(p.H - 1);
            const float kx = 2.0f / wm1, ky = 2.0f / hm1;
            for (int v = 0; v < p.V; ++v)
                cnt += project_fast(sP + (v * p.batch + c.x) * 12, X, Y, Z, wm1, hm1, kx, ky).vis ? 1 : 0;
        }
        p.count[i] = (float)cnt;
        valid = in_range && cnt >= p.min_view;
        vbatch = c.x;
    }
    const unsigned long long m = __ballot(valid);
    const int lane = tid & (kWave - 1);
    {   // per-batch valid counts: lists are grouped by batch, so a wave almost always holds one batch
        // element -> one LDS atomic per wave instead of one per voxel
        const int b0 = __shfl(vbatch, m ? (__ffsll((long long)m) - 1) : 0);
        const bool uniform = __ballot(valid && vbatch != b0) == 0ull;
        if (uniform) {
            if (m && lane == (__ffsll((long long)m) - 1)) atomicAdd(&sBatch[b0], __popcll(m));
        } else if (valid) {
            atomicAdd(&sBatch[vbatch], 1);
        }
    }
    if constexpr (VOX >= kWave) {
        // tile = VOX / 64 whole waves: per-wave popcounts through LDS
        if (lane == 0) sWave[tid / kWave] = __popcll(m);
        __syncthreads();
        constexpr int WPT = VOX / kWave;  // waves per tile
        if (tid < BLOCK / VOX) {
            int t = 0;
            for (int w = 0; w < WPT; ++w) t += sWave[tid * WPT + w];
            const int tile = blockIdx.x * (BLOCK / VOX) + tid;
            if ((long long)tile * VOX < p.n) tile_sums[tile] = t;
        }
    } else {
        if ((lane % VOX) == 0 && i < p.n) tile_sums[i / VOX] = __popcll((m >> lane) & ((1ull << VOX) - 1ull));
        __syncthreads();
    }
    // per-batch totals: per-workgroup rows reduced by bp_scan_kernel (3,456 same-address global atomics
    // cost ~40 us on the dense 96^3 level: device-scope atomics serialise at the memory side)
    if (blk_batch) {
        __syncthreads();
        for (int b = tid; b < p.batch; b += BLOCK) blk_batch[(size_t)blockIdx.x * p.batch + b] = sBatch[b];
    }
}

// exclusive scan of the block totals, one workgroup; also publishes n_valid
__global__ __launch_bounds__(1024) void bp_scan_kernel(int32_t *block_sums, int nblk,
                                                       int32_t *n_valid_dev, const int32_t *blk_batch = nullptr,
                                                       int nblk_count = 0, int batch = 0)
{
    __shared__ int sWave[1024 / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    int carry = 0;
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + tid;
        const int v = i < nblk ? block_sums[i] : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == kWave - 1) sWave[wid] = x;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 1024 / kWave; ++w) {
            const int c = sWave[w];
            woff += (w < wid) ? c : 0;
            tot += c;
        }
        if (i < nblk) block_sums[i] = carry + woff + x - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) n_valid_dev[0] = carry;
    // per-batch valid totals from the count kernel's per-workgroup rows
    if (batch == 1) {
        if (tid == 0) n_valid_dev[1] = carry;
    } else if (blk_batch) {
        for (int b = 0; b < batch; ++b) {
            int x = 0;
            for (int i = tid; i < nblk_count; i += 1024) x += blk_batch[(size_t)i * batch + b];
#pragma unroll
            for (int d = kWave / 2; d > 0; d >>= 1) x += __shfl_xor(x, d);
            __syncthreads();
            if (lane == 0) sWave[wid] = x;
            __syncthreads();
            if (tid == 0) {
                int t = 0;
                for (int w = 0; w < 1024 / kWave; ++w) t += sWave[w];
                n_valid_dev[1 + b] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2 + K3 (+K4): compaction and bilinear gather
// ---------------------------------------------------------------------------------------------
struct Taps {
    int o00, o10, o01, o11;  // element offsets of the four taps inside one NHWC map (channel 0)
    float w00, w10, w01, w11;
};

__device__ __forceinline__ Taps make_taps(float ix, float iy, int W, int H, int C)
{
    const float x0f = floorf(ix), y0f = floorf(iy);
    int x0 = (int)x0f, y0 = (int)y0f;
    float wx1 = ix - x0f, wx0 = (x0f + 1.0f) - ix;
    float wy1 = iy - y0f, wy0 = (y0f + 1.0f) - iy;
    int x1 = x0 + 1, y1 = y0 + 1;
    // zero padding: a visible voxel has ix in [0, W-1], so only the +1 taps can leave the image,
    // and then only with weight exactly 0
    if (x1 >= W) { x1 = W - 1; wx1 = 0.0f; }
    if (y1 >= H) { y1 = H - 1; wy1 = 0.0f; }
    Taps t;
    t.o00 = (y0 * W + x0) * C;
    t.o10 = (y0 * W + x1) * C;
    t.o01 = (y1 * W + x0) * C;
    t.o11 = (y1 * W + x1) * C;
    t.w00 = wx0 * wy0;
    t.w10 = wx1 * wy0;
    t.w01 = wx0 * wy1;
    t.w11 = wx1 * wy1;
    return t;
}

template <int VEC>
struct Chan;
template <>
struct Chan<4> {
    float4 v;
    __device__ __forceinline__ static Chan zero() { return Chan{make_float4(0.f, 0.f, 0.f, 0.f)}; }
    __device__ __forceinline__ static Chan sample(const float *m, const Taps &t)
    {
        const float4 a = *reinterpret_cast<const float4 *>(m + t.o00);
        const float4 b = *reinterpret_cast<const float4 *>(m + t.o10);
        const float4 c = *reinterpret_cast<const float4 *>(m + t.o01);
        const float4 d = *reinterpret_cast<const float4 *>(m + t.o11);
        Chan r;
        r.v.x = fmaf(d.x, t.w11, fmaf(c.x, t.w01, fmaf(b.x, t.w10, a.x * t.w00)));
        r.v.y = fmaf(d.y, t.w11, fmaf(c.y, t.w01, fmaf(b.y, t.w10, a.y * t.w00)));
        r.v.z = fmaf(d.z, t.w11, fmaf(c.z, t.w01, fmaf(b.z, t.w10, a.z * t.w00)));
        r.v.w = fmaf(d.w, t.w11, fmaf(c.w, t.w01, fmaf(b.w, t.w10, a.w * t.w00)));
        return r;
    }
    __device__ __forceinline__ void add(const Chan &o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
    __device__ __forceinline__ void add_sqdiff(const Chan &f, const Chan &mean)
    {
        const float dx = f.v.x - mean.v.x, dy = f.v.y - mean.v.y, dz = f.v.z - mean.v.z, dw = f.v.w - mean.v.w;
        v.x = fmaf(dx, dx, v.x); v.y = fmaf(dy, dy, v.y); v.z = fmaf(dz, dz, v.z); v.w = fmaf(dw, dw, v.w);
    }
    __device__ __forceinline__ Chan div(float d) const
    {
        return Chan{make_float4(__fdiv_rn(v.x, d), __fdiv_rn(v.y, d), __fdiv_rn(v.z, d), __fdiv_rn(v.w, d))};
    }
    __device__ __forceinline__ void store(float *dst, bool aligned16) const
    {
        if (aligned16) {
            *reinterpret_cast<float4 *>(dst) = v;
        } else {
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
};
template <>
struct Chan<1> {
    float v;
    __device__ __forceinline__ static Chan zero() { return Chan{0.f}; }
    __device__ __forceinline__ static Chan sample(const float *m, const Taps &t)
    {
        return Chan{fmaf(m[t.o11], t.w11, fmaf(m[t.o01], t.w01, fmaf(m[t.o10], t.w10, m[t.o00] * t.w00)))};
    }
    __device__ __forceinline__ void add(const Chan &o) { v += o.v; }
    __device__ __forceinline__ void add_sqdiff(const Chan &f, const Chan &mean)
    {
        const float d = f.v - mean.v;
        v = fmaf(d, d, v);
    }
    __device__ __forceinline__ Chan div(float d) const { return Chan{__fdiv_rn(v, d)}; }
    __device__ __forceinline__ void store(float *dst, bool) const { dst[0] = v; }
};

// QT > 0: channel groups per voxel known at compile time (fast div/mod); QT == 0: runtime
template <int VOX, int MODE, int VEC, int QT>
__global__ __launch_bounds__(256) void bp_gather_kernel(BpParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BLOCK = 256;
    // LDS carve (every offset a multiple of 16 bytes)
    float2 *sPix = reinterpret_cast<float2 *>(smem);                     // [VOX][V] pixel coords
    float *sP = reinterpret_cast<float *>(sPix + VOX * p.V);             // [V*B][12]
    const int nP = (p.V * p.batch * 12 + 3) & ~3;
    uint32_t *sVis = reinterpret_cast<uint32_t *>(sP + nP);              // [VOX] view bitmask
    float *sDen = reinterpret_cast<float *>(sVis + VOX);                 // [VOX] divisor
    int *sBatch = reinterpret_cast<int *>(sDen + VOX);                   // [VOX] batch index
    int *sSlot = sBatch + VOX;                                           // [VOX] rank -> thread
    int *sOut = sSlot + VOX;                                             // [VOX] thread -> output row
    int *sWave = sOut + VOX;                                             // [BLOCK/64]

    const int tid = threadIdx.x;
    const int lb = p.xcd_slabs ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    stage_matrices(sP, p.krcam, p.V * p.batch, tid, BLOCK);
    __syncthreads();

    const int e = lb * VOX + tid;
    const int i = e;
    const float wm1 = (float)(p.W - 1), hm1 = (float)(p.H - 1);
    bool valid = false;
    int4 c = make_int4(0, 0, 0, 0);
    float X = 0.f, Y = 0.f, Z = 0.f, zsum = 0.f;
    int cnt = 0;
    uint32_t vis = 0;
    if (tid < VOX && e < p.n) {
        c = reinterpret_cast<const int4 *>(p.coords)[i];
        if (c.x >= 0 && c.x < p.batch) {
            voxel_centre(c, p.origin, p.voxel_size, X, Y, Z);
            for (int v = 0; v < p.V; ++v) {
                const Proj pr = project(sP + (v * p.batch + c.x) * 12, X, Y, Z, wm1, hm1);
                // grid -> pixel exactly as grid_sample(align_corners=True) un-normalises
                const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(pr.gx, 1.0f), 2.0f), wm1);
                const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(pr.gy, 1.0f), 2.0f), hm1);
                sPix[tid * p.V + v] = make_float2(ix, iy);
                if (pr.vis) {
                    vis |= 1u << v;
                    cnt += 1;
                    zsum += pr.pz;
                }
            }
            valid = cnt >= p.min_view;
        }
    }
    int nloc;
    const int rank = block_exclusive_rank<BLOCK>(valid, sWave, nloc);
    if (nloc == 0) return;
    const int base = p.block_offsets[lb];
    const int n_valid = p.n_valid_dev[0];
    const int cout = (MODE == EPRECON_BP_MEAN_DEPTH) ? p.C + 1 : p.C;
    if (valid) {
        const int o = base + rank;
        sSlot[rank] = tid;
        sOut[tid] = o;
        sVis[tid] = vis;
        const float den = (float)(cnt > 0 ? cnt : 1);
        sDen[tid] = den;
        sBatch[tid] = c.x;
        reinterpret_cast<int4 *>(p.out_coords)[o] = c;
        if (MODE == EPRECON_BP_MEAN_DEPTH) p.out_feats[(size_t)o * cout + p.C] = __fdiv_rn(zsum, den);
        if (p.out_grid || p.out_mask) {
            for (int v = 0; v < p.V; ++v) {
                const Proj pr = project(sP + (v * p.batch + c.x) * 12, X, Y, Z, wm1, hm1);
                if (p.out_grid)
                    reinterpret_cast<float2 *>(p.out_grid)[(size_t)v * n_valid + o] = make_float2(pr.gx, pr.gy);
                if (p.out_mask) p.out_mask[(size_t)v * n_valid + o] = pr.vis ? 1 : 0;
            }
        }
    }
    __syncthreads();

    const int Q = QT > 0 ? QT : p.C / VEC;
    const size_t map_elems = (size_t)p.H * p.W * p.Cs;
    const bool aligned16 = (MODE != EPRECON_BP_MEAN_DEPTH);
    for (int w = tid; w < nloc * Q; w += BLOCK) {
        const int r = w / Q;
        const int q = w - r * Q;
        const int t = sSlot[r];
        const uint32_t vm = sVis[t];
        const float den = sDen[t];
        const int b = sBatch[t];
        const float *fb = p.feats_nhwc + (size_t)b * map_elems + q * VEC;
        const size_t vstride = (size_t)p.batch * map_elems;
        Chan<VEC> acc = Chan<VEC>::zero();
        for (int v = 0; v < p.V; ++v) {
            if (vm & (1u << v)) {
                const float2 px = sPix[t * p.V + v];
                const Taps tp = make_taps(px.x, px.y, p.W, p.H, p.Cs);
                acc.add(Chan<VEC>::sample(fb + (size_t)v * vstride, tp));
            }
        }
        const int orow = sOut[t];
        float *dst = p.out_feats + (size_t)orow * cout + q * VEC;
        if (MODE == EPRECON_BP_VARIANCE) {
            // models/occupancy_initialization.py:127-128: population variance over visible views
            const Chan<VEC> mean = acc.div(den);
            Chan<VEC> sq = Chan<VEC>::zero();
            for (int v = 0; v < p.V; ++v) {
                if (vm & (1u << v)) {
                    const float2 px = sPix[t * p.V + v];
                    const Taps tp = make_taps(px.x, px.y, p.W, p.H, p.Cs);
                    sq.add_sqdiff(Chan<VEC>::sample(fb + (size_t)v * vstride, tp), mean);
                }
            }
            sq.div(den).store(dst, true);
            if (p.out_mean) mean.store(p.out_mean + (size_t)orow * p.C + q * VEC, true);
        } else {
            acc.div(den).store(dst, aligned16);
        }
    }
}


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// K2 + K3 (+K4), default for C % 4 == 0: direct gather with per-pair taps in LDS and U views of loads
// in flight per lane (U = 1 by default: measured, the kernel is bound by the L1 access rate and extra
// loads in flight only cost registers).
//   phase 1a  one thread per (voxel, view): cheap bit-exact projection; byte offset of tap (x0, y0)
//             (base pixel clamped to [0, W-2] x [0, H-2], so all four taps are in the image and the
//             border weight is exactly 0 where the reference pads with zeros) and the two fractional
//             weights -> LDS, 12 bytes per pair, computed once instead of once per channel group;
//   phase 1b  one thread per voxel: visible count, validity, stable in-tile compaction, output row;
//   phase 2   one thread per (valid voxel, 4 channels): the visible views are walked U at a time:
//             4 U buffer loads (32-bit offsets; the other three taps are scalar offsets of the
//             first) are issued before the first fma; slots past the last visible view point out of
//             the buffer range, which costs no memory traffic.  Views accumulate in ascending order.
// ---------------------------------------------------------------------------------------------
constexpr int kOobOffset = (int)0x80000000u;  // >= num_records (maps are limited to < 2 GiB here)

template <int U, class F>
__device__ __forceinline__ void visit_visible_samples(__amdgpu_buffer_rsrc_t rsrc, const int *sOffRow,
                                                      const float2 *sWxyRow, uint32_t vm, int qbyte, int s10,
                                                      int s01, int s11, F &&f)
{
    uint32_t m = vm;
    while (m) {
        int off[U];
        float2 wxy[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            on[u] = m != 0;
            const int v = on[u] ? __builtin_ctz(m) : 0;
            m &= m - 1;
            wxy[u] = sWxyRow[v];
            off[u] = on[u] ? sOffRow[v] + qbyte : kOobOffset;
        }
        u32x4 a[U], b[U], c[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[u], 0, 0);
            b[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[u], s10, 0);
            c[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[u], s01, 0);
            d[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[u], s11, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float wx1 = wxy[u].x, wy1 = wxy[u].y;
            const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;  // exact: wx1 is a multiple of ulp(ix) in [0, 1]
            const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
            float4 r;
#define EP_TAP(k) fmaf(__uint_as_float(d[u].k), w11, fmaf(__uint_as_float(c[u].k), w01, fmaf(__uint_as_float(b[u].k), w10, __uint_as_float(a[u].k) * w00)))
            r.x = EP_TAP(x); r.y = EP_TAP(y); r.z = EP_TAP(z); r.w = EP_TAP(w);
#undef EP_TAP
            if (on[u]) f(r);
        }
    }
}

template <int VOX, int MODE, int QT, int U>
__global__ __launch_bounds__(256) void bp_gather_mlp_kernel(BpParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BLOCK = 256;
    constexpr int VPT = BLOCK / VOX;  // threads per voxel in phase 1a (1, 4 or 16)
    float2 *sWxy = reinterpret_cast<float2 *>(smem);                      // [VOX*V] fractional weights
    int *sOff = reinterpret_cast<int *>(sWxy + VOX * p.V);                // [VOX*V] byte offset of tap (x0, y0)
    float *sP = reinterpret_cast<float *>(sOff + VOX * p.V);              // [V*B][12]
    const int nP = (p.V * p.batch * 12 + 3) & ~3;
    uint32_t *sVis = reinterpret_cast<uint32_t *>(sP + nP);               // [VOX] view bitmask
    int *sSlot = reinterpret_cast<int *>(sVis + VOX);                     // [VOX] rank -> voxel
    int *sOut = sSlot + VOX;                                              // [VOX] voxel -> output row
    int *sWave = sOut + VOX;                                              // [BLOCK/64]

    const int tid = threadIdx.x;
    const int lb = p.xcd_slabs ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    stage_matrices(sP, p.krcam, p.V * p.batch, tid, BLOCK);
    if (VPT > 1 && tid < VOX) sVis[tid] = 0;
    __syncthreads();

    const float wm1 = (float)(p.W - 1), hm1 = (float)(p.H - 1);
    const float kx = 2.0f / wm1, ky = 2.0f / hm1;
    const int map_elems = p.H * p.W * p.Cs;
    // ---- phase 1a: thread -> (voxel tid % VOX, views tid / VOX, + VPT, ...) ----
    const int vx = tid % VOX;
    const int e = lb * VOX + vx;
    int4 c = make_int4(-1, 0, 0, 0);
    if (e < p.n) c = reinterpret_cast<const int4 *>(p.coords)[e];
    const bool in_batch = c.x >= 0 && c.x < p.batch;
    uint32_t vis = 0;
    if (in_batch) {
        float X, Y, Z;
        voxel_centre(c, p.origin, p.voxel_size, X, Y, Z);
        for (int v = tid / VOX; v < p.V; v += VPT) {
            const ProjFast q = project_fast(sP + (v * p.batch + c.x) * 12, X, Y, Z, wm1, hm1, kx, ky);
            if (!q.vis) continue;
            const float x0f = fminf(floorf(q.u), wm1 - 1.0f), y0f = fminf(floorf(q.v), hm1 - 1.0f);
            sWxy[vx * p.V + v] = make_float2(q.u - x0f, q.v - y0f);
            sOff[vx * p.V + v] = ((v * p.batch + c.x) * map_elems + ((int)y0f * p.W + (int)x0f) * p.Cs) * 4;
            vis |= 1u << v;
        }
        if (VPT > 1 && vis) atomicOr(&sVis[vx], vis);
    }
    if (VPT > 1) {
        __syncthreads();
        vis = sVis[vx];
    }
    // ---- phase 1b ----
    const int cnt = __popc(vis);
    const bool valid = tid < VOX && in_batch && cnt >= p.min_view;
    int nloc;
    const int rank = block_exclusive_rank<BLOCK>(valid, sWave, nloc);
    if (nloc == 0) return;
    const int n_valid = p.n_valid_dev[0];
    const int cout = (MODE == EPRECON_BP_MEAN_DEPTH) ? p.C + 1 : p.C;
    if (valid) {
        const int o = p.block_offsets[lb] + rank;
        sSlot[rank] = tid;
        sOut[tid] = o;
        if (VPT == 1) sVis[tid] = vis;
        reinterpret_cast<int4 *>(p.out_coords)[o] = c;
        if (MODE == EPRECON_BP_MEAN_DEPTH || p.out_grid || p.out_mask) {
            float X, Y, Z, zsum = 0.0f;
            voxel_centre(c, p.origin, p.voxel_size, X, Y, Z);
            for (int v = 0; v < p.V; ++v) {
                const Proj pr = project(sP + (v * p.batch + c.x) * 12, X, Y, Z, wm1, hm1);
                if (pr.vis) zsum += pr.pz;
                if (p.out_grid)
                    reinterpret_cast<float2 *>(p.out_grid)[(size_t)v * n_valid + o] = make_float2(pr.gx, pr.gy);
                if (p.out_mask) p.out_mask[(size_t)v * n_valid + o] = pr.vis ? 1 : 0;
            }
            if (MODE == EPRECON_BP_MEAN_DEPTH)
                p.out_feats[(size_t)o * cout + p.C] = __fdiv_rn(zsum, (float)(cnt > 0 ? cnt : 1));
        }
    }
    __syncthreads();
    // ---- phase 2 ----
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.feats_nhwc), 0, p.V * p.batch * map_elems * 4, 0x00020000);
    const int Q = QT > 0 ? QT : p.C / 4;
    const int s10 = p.Cs * 4, s01 = p.W * p.Cs * 4, s11 = s01 + s10;
    for (int w = tid; w < nloc * Q; w += BLOCK) {
        const int r = w / Q, q = w - r * Q;
        const int t = sSlot[r];
        const uint32_t vm = sVis[t];
        const float den = (float)max(__popc(vm), 1);
        const int *offRow = sOff + t * p.V;
        const float2 *wxyRow = sWxy + t * p.V;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        visit_visible_samples<U>(rsrc, offRow, wxyRow, vm, q * 16, s10, s01, s11, [&](const float4 &f) {
            acc.x += f.x; acc.y += f.y; acc.z += f.z; acc.w += f.w;
        });
        const int orow = sOut[t];
        float *dst = p.out_feats + (size_t)orow * cout + q * 4;
        const float4 mean = make_float4(__fdiv_rn(acc.x, den), __fdiv_rn(acc.y, den), __fdiv_rn(acc.z, den),
                                        __fdiv_rn(acc.w, den));
        if (MODE == EPRECON_BP_VARIANCE) {
            // models/occupancy_initialization.py:127-128: population variance over visible views
            float4 sq = make_float4(0.f, 0.f, 0.f, 0.f);
            visit_visible_samples<U>(rsrc, offRow, wxyRow, vm, q * 16, s10, s01, s11, [&](const float4 &f) {
                const float dx = f.x - mean.x, dy = f.y - mean.y, dz = f.z - mean.z, dw = f.w - mean.w;
                sq.x = fmaf(dx, dx, sq.x); sq.y = fmaf(dy, dy, sq.y); sq.z = fmaf(dz, dz, sq.z); sq.w = fmaf(dw, dw, sq.w);
            });
            *reinterpret_cast<float4 *>(dst) = make_float4(__fdiv_rn(sq.x, den), __fdiv_rn(sq.y, den),
                                                           __fdiv_rn(sq.z, den), __fdiv_rn(sq.w, den));
            if (p.out_mean) *reinterpret_cast<float4 *>(p.out_mean + (size_t)orow * p.C + q * 4) = mean;
        } else if (MODE == EPRECON_BP_MEAN_DEPTH) {
            dst[0] = mean.x; dst[1] = mean.y; dst[2] = mean.z; dst[3] = mean.w;  // rows of C + 1 floats
        } else {
            *reinterpret_cast<float4 *>(dst) = mean;
        }
    }
}


// ops/back_project.py:69-75 — per batch element: mu = mean(d[d>0]); sigma = ||d[d>0]-mu||_2 + 1e-5;
// d_hat = (d-mu)/sigma, 0 where d <= 0.  One workgroup per batch element, three sweeps.
__global__ __launch_bounds__(1024) void bp_depth_norm_kernel(float *out_feats, int cout,
                                                             const int32_t *n_valid_dev)
{
    __shared__ float sRed[1024 / kWave];
    __shared__ float sBcast[2];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    int start = 0;
    for (int k = 0; k < b; ++k) start += n_valid_dev[1 + k];
    const int len = n_valid_dev[1 + b];
    float *d = out_feats + (size_t)start * cout + (cout - 1);

    auto block_sum = [&](float x) -> float {
#pragma unroll
        for (int s = kWave / 2; s > 0; s >>= 1) x += __shfl_xor(x, s);
        __syncthreads();
        if (lane == 0) sRed[wid] = x;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 1024 / kWave; ++w) t += sRed[w];
        return t;
    };
    float s = 0.f, m = 0.f;
    for (int j = tid; j < len; j += 1024) {
        const float x = d[(size_t)j * cout];
        if (x > 0.f) { s += x; m += 1.f; }
    }
    const float tot = block_sum(s), cntp = block_sum(m);
    const float mu = tot / cntp;
    float ss = 0.f;
    for (int j = tid; j < len; j += 1024) {
        const float x = d[(size_t)j * cout];
        if (x > 0.f) ss = fmaf(x - mu, x - mu, ss);
    }
    const float sigma = sqrtf(block_sum(ss)) + 1e-5f;
    for (int j = tid; j < len; j += 1024) {
        const float x = d[(size_t)j * cout];
        d[(size_t)j * cout] = x > 0.f ? (x - mu) / sigma : 0.f;
    }
    (void)sBcast;
}

// ---------------------------------------------------------------------------------------------
// NCHW -> NHWC through an LDS tile: reads coalesced along H*W, writes coalesced along (pixel, C)
// ---------------------------------------------------------------------------------------------
constexpr int kTrPix = 64;
// zero / zero_n (optional): int32 words the first block clears on its way — the valid-voxel counters of the back-projection
// this re-layout is the first launch of (one launch less than a memset in front of it)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ in,
                                                           float *__restrict__ out, int C, int hw, int Cs,
                                                           int32_t *zero = nullptr, int zero_n = 0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);  // [C][kTrPix + 1]
    if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < zero_n) zero[threadIdx.x] = 0;
    const int map = blockIdx.y;
    const int p0 = blockIdx.x * kTrPix;
    const int npix = min(kTrPix, hw - p0);
    const float *src = in + (size_t)map * C * hw;
    float *dst = out + (size_t)map * hw * Cs + (size_t)p0 * Cs;
    for (int e = threadIdx.x; e < C * kTrPix; e += 256) {
        const int c = e / kTrPix, px = e - c * kTrPix;
        if (px < npix) tile[c * (kTrPix + 1) + px] = src[(size_t)c * hw + p0 + px];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < npix * Cs; e += 256) {  // pad channels (Cs > C) are written as zeros
        const int px = e / Cs, c = e - px * Cs;
        dst[e] = c < C ? tile[c * (kTrPix + 1) + px] : 0.0f;
    }
}


// the per-view maps of up to three levels in one launch (eprecon_views_to_rows_async): block -> (level, view, 64-pixel tile)
struct ViewsParams {
    eprecon_views_desc d;
    int tile0[4];        // first block of level l; tile0[levels] = grid size
    int tiles[3];        // 64-pixel tiles per map of level l
};
__global__ __launch_bounds__(256) void views_to_rows_kernel(ViewsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);  // [C][kTrPix + 1]
    const int b = blockIdx.x;
    const int l = b >= p.tile0[2] ? 2 : (b >= p.tile0[1] ? 1 : 0);
    const int r = b - p.tile0[l];
    const int v = r / p.tiles[l], t = r - v * p.tiles[l];
    const int C = p.d.channels[l], hw = p.d.hw[l];
    const int p0 = t * kTrPix;
    const int npix = min(kTrPix, hw - p0);
    const float *src = p.d.src[l][v];
    float *dst = p.d.dst[l] + ((size_t)v * hw + p0) * C;
    for (int e = threadIdx.x; e < C * kTrPix; e += 256) {
        const int c = e / kTrPix, px = e - c * kTrPix;
        if (px < npix) tile[c * (kTrPix + 1) + px] = src[(size_t)c * hw + p0 + px];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < npix * C; e += 256) {
        const int px = e / C, c = e - px * C;
        dst[e] = tile[c * (kTrPix + 1) + px];
    }
}

// ---------------------------------------------------------------------------------------------
// Backward of the back-projection with respect to the image features (training): the
// transpose of the bilinear gather is a scatter of four weighted taps per (voxel, visible view, channel).
//   MEAN / MEAN_DEPTH   d f_v = d out / cnt                          (the mean-depth channel has no feature gradient)
//   VARIANCE            d f_v = 2 (f_v - mean) / cnt * d var + d mean / cnt
// One thread per (valid voxel, channel): consecutive lanes hit consecutive addresses of one pixel, so the
// hardware float atomics of a wave coalesce.  Same projection and tap arithmetic as bp_gather_kernel.
// ---------------------------------------------------------------------------------------------
struct BpBwdParams {
    const int32_t *coords; int64_t n;      // the VALID voxels (out_coords of the forward)
    const float *origin; int batch; float voxel_size;
    const float *feats_nhwc; const float *krcam;
    int V, C, H, W, mode;
    const float *dout; int ld_dout;
    const float *dmean;                    // VARIANCE only, may be null
    float *dfeats;                         // [V*B][H*W][C], zeroed by the caller
    unsigned long long *dfix;              // the same elements as 64-bit fixed point (deterministic form), or null
};

constexpr double kFixScale = 1099511627776.0;   // 2^40
__device__ __forceinline__ unsigned long long to_fixed(float v)
{
    const double x = fmin(fmax((double)v * kFixScale, -9.0e18), 9.0e18);
    return (unsigned long long)__double2ll_rn(x);           // two's complement: an unsigned add is a signed add
}
// A non-finite contribution (a diverging training run) must stay visible: the clamp above would turn NaN / Inf into a finite
// +-8.2e6.  Such a contribution is not added; the element of the fp32 output (zeroed by the caller) is marked NaN instead — a
// plain store of one value, so still independent of the order — and the conversion below leaves marked elements alone.
__device__ __forceinline__ void fixed_add(unsigned long long *acc, float *mark, float v)
{
    if (__builtin_isfinite(v)) atomicAdd(acc, to_fixed(v));
    else *mark = __builtin_nanf("");
}
__global__ void fixed_to_float_kernel(const unsigned long long *acc, long long n, float *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !__builtin_isnan(out[i])) out[i] = (float)((double)(long long)acc[i] * (1.0 / kFixScale));
}

__global__ __launch_bounds__(256) void bp_backward_kernel(BpBwdParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sP = reinterpret_cast<float *>(smem);
    stage_matrices(sP, p.krcam, p.V * p.batch, threadIdx.x, 256);
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= p.n * p.C) return;
    const int64_t i = e / p.C;
    const int ch = (int)(e - i * p.C);
    const int4 c = reinterpret_cast<const int4 *>(p.coords)[i];
    if (c.x < 0 || c.x >= p.batch) return;
    float X, Y, Z;
    voxel_centre(c, p.origin, p.voxel_size, X, Y, Z);
    const float wm1 = (float)(p.W - 1), hm1 = (float)(p.H - 1);
    const size_t map_elems = (size_t)p.H * p.W * p.C;
    int cnt = 0;
    for (int v = 0; v < p.V; ++v) cnt += project(sP + (v * p.batch + c.x) * 12, X, Y, Z, wm1, hm1).vis ? 1 : 0;
    if (cnt == 0) return;
    const float inv = 1.0f / (float)cnt;
    const float g = p.dout[i * p.ld_dout + ch];
    float mean = 0.0f;
    if (p.mode == EPRECON_BP_VARIANCE) {
        for (int v = 0; v < p.V; ++v) {
            const Proj pr = project(sP + (v * p.batch + c.x) * 12, X, Y, Z, wm1, hm1);
            if (!pr.vis) continue;
            const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(pr.gx, 1.0f), 2.0f), wm1);
            const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(pr.gy, 1.0f), 2.0f), hm1);
            mean += Chan<1>::sample(p.feats_nhwc + ((size_t)v * p.batch + c.x) * map_elems + ch, make_taps(ix, iy, p.W, p.H, p.C)).v;
        }
        mean *= inv;
    }
    const float gm = (p.mode == EPRECON_BP_VARIANCE && p.dmean) ? p.dmean[i * p.C + ch] * inv : 0.0f;
    for (int v = 0; v < p.V; ++v) {
        const Proj pr = project(sP + (v * p.batch + c.x) * 12, X, Y, Z, wm1, hm1);
        if (!pr.vis) continue;
        const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(pr.gx, 1.0f), 2.0f), wm1);
        const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(pr.gy, 1.0f), 2.0f), hm1);
        const Taps t = make_taps(ix, iy, p.W, p.H, p.C);
        const size_t mo = ((size_t)v * p.batch + c.x) * map_elems + ch;
        float gv;
        if (p.mode == EPRECON_BP_VARIANCE) {
            const float f = Chan<1>::sample(p.feats_nhwc + mo, t).v;
            gv = 2.0f * (f - mean) * inv * g + gm;
        } else {
            gv = g * inv;
        }
        if (p.dfix) {
            // order-independent accumulation: 64-bit fixed point (2^-40 resolution, |sum| < 8.4e6), integer atomics
            unsigned long long *d = p.dfix + mo;
            float *m = p.dfeats + mo;
            if (t.w00 != 0.0f) fixed_add(d + t.o00, m + t.o00, t.w00 * gv);
            if (t.w10 != 0.0f) fixed_add(d + t.o10, m + t.o10, t.w10 * gv);
            if (t.w01 != 0.0f) fixed_add(d + t.o01, m + t.o01, t.w01 * gv);
            if (t.w11 != 0.0f) fixed_add(d + t.o11, m + t.o11, t.w11 * gv);
        } else {
            float *d = p.dfeats + mo;
            if (t.w00 != 0.0f) unsafeAtomicAdd(d + t.o00, t.w00 * gv);
            if (t.w10 != 0.0f) unsafeAtomicAdd(d + t.o10, t.w10 * gv);
            if (t.w01 != 0.0f) unsafeAtomicAdd(d + t.o01, t.w01 * gv);
            if (t.w11 != 0.0f) unsafeAtomicAdd(d + t.o11, t.w11 * gv);
        }
    }
}

struct ProfileState {
    bool on = false, recorded = false, one_shot = false;
    hipEvent_t start = nullptr, stop = nullptr;
    const char *kernel = "";
} g_prof;

size_t gather_lds_bytes(int vox, int V, int B)
{
    const size_t nP = ((size_t)V * B * 12 + 3) & ~(size_t)3;
    return (size_t)vox * V * sizeof(float2) + nP * sizeof(float) + (size_t)vox * 5 * 4 +
           (size_t)(256 / kWave) * 4 + 16;
}

template <int VOX, int MODE>
int launch_gather(const BpParams &p, int nblk, hipStream_t st)
{
    const size_t lds = gather_lds_bytes(VOX, p.V, p.batch);
    const dim3 grid(nblk), block(256);
#define EP_GATHER(VEC, QT) hipLaunchKernelGGL((bp_gather_kernel<VOX, MODE, VEC, QT>), grid, block, lds, st, p)
    if (p.C % 4 == 0) {
        switch (p.C / 4) {
            case 6: EP_GATHER(4, 6); break;    // C = 24  (1/4-res level)
            case 8: EP_GATHER(4, 8); break;    // C = 32  (fused initialisation maps)
            case 10: EP_GATHER(4, 10); break;  // C = 40  (1/8-res level)
            case 20: EP_GATHER(4, 20); break;  // C = 80  (1/16-res level)
            default: EP_GATHER(4, 0); break;
        }
    } else {
        EP_GATHER(1, 0);
    }
#undef EP_GATHER
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

size_t gather_mlp_lds_bytes(int vox, int V, int B)
{
    const size_t nP = ((size_t)V * B * 12 + 3) & ~(size_t)3;
    return (size_t)vox * V * 12 + nP * sizeof(float) + (size_t)vox * 3 * 4 + (size_t)(256 / kWave) * 4 + 16;
}

bool gather_mlp_supported(const BpParams &p)
{
    return p.C % 4 == 0 && p.Cs % 4 == 0 && p.V <= 32 &&
           (size_t)p.V * p.batch * p.H * p.W * p.Cs * 4 < 0x7fff0000ull &&
           gather_mlp_lds_bytes(256, p.V, p.batch) <= 64 * 1024;
}

template <int VOX, int MODE, int U>
int launch_gather_mlp_u(const BpParams &p, int nblk, hipStream_t st)
{
    const size_t lds = gather_mlp_lds_bytes(VOX, p.V, p.batch);
    const dim3 grid(nblk), block(256);
#define EP_GATHER(QT) hipLaunchKernelGGL((bp_gather_mlp_kernel<VOX, MODE, QT, U>), grid, block, lds, st, p)
    switch (p.C / 4) {
        case 6: EP_GATHER(6); break;    // C = 24  (1/4-res level)
        case 8: EP_GATHER(8); break;    // C = 32  (fused initialisation maps)
        case 10: EP_GATHER(10); break;  // C = 40  (1/8-res level)
        case 20: EP_GATHER(20); break;  // C = 80  (1/16-res level)
        default: EP_GATHER(0); break;
    }
#undef EP_GATHER
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

template <int VOX, int MODE>
int launch_gather_mlp(const BpParams &p, int nblk, hipStream_t st)
{
    return launch_gather_mlp_u<VOX, MODE, 1>(p, nblk, st);   // (2 / 3 / 4 views of loads in flight per lane: slower, see above)
}

}  // namespace

namespace ep {
int profile_bracket_begin(hipStream_t st)
{
    if (g_prof.on && g_prof.start) EP_HIP_CHECK(hipEventRecord(g_prof.start, st));
    return EPRECON_OK;
}
int profile_bracket_end(hipStream_t st, const char *kernel)
{
    if (g_prof.on && g_prof.start) {
        EP_HIP_CHECK(hipEventRecord(g_prof.stop, st));
        g_prof.recorded = true;
        g_prof.kernel = kernel;
        if (g_prof.one_shot) g_prof.on = false;
    }
    return EPRECON_OK;
}
}  // namespace ep

extern "C" {

const char *eprecon_profile_gather_kernel(void) { return g_prof.kernel; }

int eprecon_abi_version(void) { return EPRECON_ABI_VERSION; }
const char *eprecon_build_arch(void) { return "gfx950"; }

size_t eprecon_back_project_workspace_bytes(int64_t n, int batch, int n_views, int channels,
                                            int height, int width, int feats_layout)
{
    size_t bytes = ep::align_up((size_t)ep::ceil_div(n > 0 ? n : 1, 16) * sizeof(int32_t), 256);
    bytes += ep::align_up((size_t)ep::ceil_div(n > 0 ? n : 1, 256) * (batch > 0 ? batch : 1) * sizeof(int32_t), 256);
    if (feats_layout == EPRECON_LAYOUT_NCHW)
        bytes += ep::align_up((size_t)n_views * batch * channels * height * width * sizeof(float), 256);
    return bytes + 256;
}

int eprecon_profile_enable(int on)
{
    if (on && !g_prof.start) {
        EP_HIP_CHECK(hipEventCreate(&g_prof.start));
        EP_HIP_CHECK(hipEventCreate(&g_prof.stop));
    }
    g_prof.on = on != 0;
    g_prof.one_shot = on == 2;
    if (on != 0) g_prof.recorded = false;  // disabling keeps the last recorded pair readable
    return EPRECON_OK;
}

float eprecon_profile_gather_ms(void)
{
    if (!g_prof.recorded) return -1.0f;
    if (hipEventSynchronize(g_prof.stop) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, g_prof.start, g_prof.stop) != hipSuccess) return -1.0f;
    return ms;
}

int eprecon_nchw_to_nhwc_async(const float *in, float *out, int maps, int channels, int hw,
                               void *stream)
{
    if (!in || !out || maps <= 0 || channels <= 0 || hw <= 0) return EPRECON_ERR_ARG;
    const size_t lds = (size_t)channels * (kTrPix + 1) * sizeof(float);
    if (lds > 64 * 1024) return EPRECON_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)ep::ceil_div(hw, kTrPix), (unsigned)maps);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), lds, (hipStream_t)stream, in, out,
                       channels, hw, channels);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_views_to_rows_async(const eprecon_views_desc *desc, void *stream)
{
    if (!desc || desc->levels < 1 || desc->levels > 3 || desc->n_views < 1 || desc->n_views > 16) return EPRECON_ERR_ARG;
    ViewsParams p = {};
    p.d = *desc;
    int cmax = 0, at = 0;
    for (int l = 0; l < 3; ++l) {
        p.tile0[l] = at;
        if (l >= desc->levels) { p.tiles[l] = 1; continue; }
        if (desc->channels[l] <= 0 || desc->hw[l] <= 0 || !desc->dst[l]) return EPRECON_ERR_ARG;
        for (int v = 0; v < desc->n_views; ++v)
            if (!desc->src[l][v]) return EPRECON_ERR_ARG;
        p.tiles[l] = ep::ceil_div(desc->hw[l], kTrPix);
        at += p.tiles[l] * desc->n_views;
        cmax = desc->channels[l] > cmax ? desc->channels[l] : cmax;
    }
    p.tile0[3] = at;
    for (int l = desc->levels; l < 3; ++l) p.tile0[l] = at;      // (no block maps to an absent level)
    const size_t lds = (size_t)cmax * (kTrPix + 1) * sizeof(float);
    if (lds > 64 * 1024) return EPRECON_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(views_to_rows_kernel, dim3((unsigned)at), dim3(256), lds, (hipStream_t)stream, p);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_back_project_async(const int32_t *coords, int64_t n, const float *origin, int batch,
                               float voxel_size, const float *feats, int feats_layout,
                               const float *krcam, int n_views, int channels, int height,
                               int width, int min_view, int mode, float *out_feats,
                               float *out_mean, int32_t *out_coords, float *count,
                               float *out_grid, uint8_t *out_mask, int32_t *n_valid_dev,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || n > 0x7fffffff / 64 || batch <= 0 || n_views <= 0 || n_views > 32 || channels <= 0 ||
        height <= 1 || width <= 1 || mode < 0 || mode > 2)
        return EPRECON_ERR_ARG;
    if (!origin || !feats || !krcam || !n_valid_dev || !workspace) return EPRECON_ERR_ARG;
    if (n > 0 && (!coords || !out_feats || !out_coords || !count)) return EPRECON_ERR_ARG;
    if ((size_t)n_views * batch * 12 * sizeof(float) > 32 * 1024) return EPRECON_ERR_UNSUPPORTED;
    if (workspace_bytes < eprecon_back_project_workspace_bytes(n, batch, n_views, channels, height,
                                                               width, feats_layout))
        return EPRECON_ERR_WORKSPACE;
    if ((size_t)n_views * batch * channels * height * width > 0x7fffffffull) return EPRECON_ERR_UNSUPPORTED;

    // (the counters are cleared by the re-layout launch below when there is one: NCHW features, <= 255 batch elements)
    const bool clear_in_relayout = n > 0 && feats_layout == EPRECON_LAYOUT_NCHW && batch < 256;
    if (!clear_in_relayout) EP_HIP_CHECK(hipMemsetAsync(n_valid_dev, 0, sizeof(int32_t) * (size_t)(1 + batch), st));
    if (n == 0) return EPRECON_OK;

    char *ws = reinterpret_cast<char *>(workspace);
    int32_t *block_sums = reinterpret_cast<int32_t *>(ws);
    ws += ep::align_up((size_t)ep::ceil_div(n, 16) * sizeof(int32_t), 256);
    int32_t *blk_batch = reinterpret_cast<int32_t *>(ws);
    ws += ep::align_up((size_t)ep::ceil_div(n, 256) * batch * sizeof(int32_t), 256);
    const float *nhwc = feats;
    int pix_stride = channels;
    if (feats_layout == EPRECON_LAYOUT_NCHW) {
        float *tmp = reinterpret_cast<float *>(ws);
        const size_t lds_t = (size_t)channels * (kTrPix + 1) * sizeof(float);
        if (lds_t > 64 * 1024) return EPRECON_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)ep::ceil_div(height * width, kTrPix), (unsigned)(n_views * batch)),
                           dim3(256), lds_t, st, feats, tmp, channels, height * width, pix_stride, n_valid_dev, 1 + batch);
        EP_LAUNCH_CHECK();
        nhwc = tmp;
        ws += ep::align_up((size_t)n_views * batch * pix_stride * height * width * sizeof(float), 256);
    } else if (feats_layout != EPRECON_LAYOUT_NHWC) {
        return EPRECON_ERR_ARG;
    }

    BpParams p;
    p.coords = coords; p.n = (int)n; p.origin = origin; p.batch = batch; p.voxel_size = voxel_size;
    p.feats_nhwc = nhwc; p.krcam = krcam; p.V = n_views; p.C = channels; p.Cs = pix_stride; p.H = height; p.W = width;
    p.min_view = min_view; p.out_feats = out_feats; p.out_mean = out_mean; p.out_coords = out_coords;
    p.count = count; p.out_grid = out_grid; p.out_mask = out_mask; p.n_valid_dev = n_valid_dev;
    p.block_offsets = block_sums;
    {   // EPRECON_BP_XCD_SLABS=0 (read per call): the gather's tiles in hardware block order — round-robin over the eight XCDs, so
        // every XCD's L2 sees tiles from the whole volume — instead of one contiguous slab of the raster per XCD.  Same results.
        const char *e = getenv("EPRECON_BP_XCD_SLABS");
        p.xcd_slabs = (e && e[0] == '0') ? 0 : 1;
    }

    // Tile = voxels handed to one 256-thread workgroup of the gather kernel.  Short lists get
    // small tiles so that the launch still covers the 256 CUs with several waves each
    // (13,824 voxels -> 864 workgroups of 16; 110,592 -> 1,728 of 64).
    const int vox = n >= 512 * 1024 ? 256 : (n >= 48 * 1024 ? 64 : 16);
    const int ntile = (int)ep::ceil_div(n, vox);
    const int nblk_count = (int)ep::ceil_div(n, 256);
    const size_t lds_count = ((size_t)n_views * batch * 12 + batch + 256 / ep::kWave) * 4 + 16;
    int32_t *bb = batch > 1 ? blk_batch : nullptr;
    if (vox == 256)
        hipLaunchKernelGGL((bp_count_kernel<256>), dim3(nblk_count), dim3(256), lds_count, st, p, block_sums, bb);
    else if (vox == 64)
        hipLaunchKernelGGL((bp_count_kernel<64>), dim3(nblk_count), dim3(256), lds_count, st, p, block_sums, bb);
    else
        hipLaunchKernelGGL((bp_count_kernel<16>), dim3(nblk_count), dim3(256), lds_count, st, p, block_sums, bb);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bp_scan_kernel, dim3(1), dim3(1024), 0, st, block_sums, ntile, n_valid_dev,
                       (const int32_t *)bb, nblk_count, batch);
    EP_LAUNCH_CHECK();

    const bool prof = g_prof.on && g_prof.start;
    if (prof) EP_HIP_CHECK(hipEventRecord(g_prof.start, st));
    int rc;
#define EP_MODE_DISPATCH(VOX)                                                                        \
    rc = mode == EPRECON_BP_MEAN ? launch_gather<VOX, EPRECON_BP_MEAN>(p, ntile, st)                   \
       : mode == EPRECON_BP_MEAN_DEPTH ? launch_gather<VOX, EPRECON_BP_MEAN_DEPTH>(p, ntile, st)       \
                                       : launch_gather<VOX, EPRECON_BP_VARIANCE>(p, ntile, st)
#define EP_MODE_DISPATCH_MLP(VOX)                                                                    \
    rc = mode == EPRECON_BP_MEAN ? launch_gather_mlp<VOX, EPRECON_BP_MEAN>(p, ntile, st)               \
       : mode == EPRECON_BP_MEAN_DEPTH ? launch_gather_mlp<VOX, EPRECON_BP_MEAN_DEPTH>(p, ntile, st)   \
                                       : launch_gather_mlp<VOX, EPRECON_BP_VARIANCE>(p, ntile, st)
    if (gather_mlp_supported(p)) {
        if (vox == 256) { EP_MODE_DISPATCH_MLP(256); }
        else if (vox == 64) { EP_MODE_DISPATCH_MLP(64); }
        else { EP_MODE_DISPATCH_MLP(16); }
    }
    else if (vox == 256) { EP_MODE_DISPATCH(256); }
    else if (vox == 64) { EP_MODE_DISPATCH(64); }
    else { EP_MODE_DISPATCH(16); }
#undef EP_MODE_DISPATCH
#undef EP_MODE_DISPATCH_MLP
    if (rc != EPRECON_OK) return rc;
    if (prof) {
        EP_HIP_CHECK(hipEventRecord(g_prof.stop, st));
        g_prof.recorded = true;
        g_prof.kernel = gather_mlp_supported(p) ? "bp_gather_mlp_kernel" : "bp_gather_kernel";
        if (g_prof.one_shot) g_prof.on = false;
    }
    if (mode == EPRECON_BP_MEAN_DEPTH) {
        hipLaunchKernelGGL(bp_depth_norm_kernel, dim3(batch), dim3(1024), 0, st, out_feats,
                           channels + 1, n_valid_dev);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}

int eprecon_back_project(const int32_t *coords, int64_t n, const float *origin, int batch,
                         float voxel_size, const float *feats, int feats_layout,
                         const float *krcam, int n_views, int channels, int height, int width,
                         int min_view, int mode, int min_valid_per_batch, float *out_feats,
                         float *out_mean, int32_t *out_coords, float *count, float *out_grid,
                         uint8_t *out_mask, int32_t *n_valid_dev, int32_t *n_valid_host,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    if (!n_valid_host) return EPRECON_ERR_ARG;
    const int rc = eprecon_back_project_async(coords, n, origin, batch, voxel_size, feats, feats_layout,
                                              krcam, n_views, channels, height, width, min_view, mode,
                                              out_feats, out_mean, out_coords, count, out_grid, out_mask,
                                              n_valid_dev, workspace, workspace_bytes, stream);
    if (rc != EPRECON_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    EP_HIP_CHECK(hipMemcpyAsync(n_valid_host, n_valid_dev, sizeof(int32_t) * (size_t)(1 + batch),
                                hipMemcpyDeviceToHost, st));
    EP_HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < batch; ++b)
        if (n_valid_host[1 + b] < min_valid_per_batch) return EPRECON_EMPTY;
    return EPRECON_OK;
}

static int bp_backward_impl(const int32_t *coords_valid, int64_t n_valid, const float *origin, int batch, float voxel_size,
                            const float *feats_nhwc, const float *krcam, int n_views, int channels, int height, int width, int mode,
                            const float *dout, int ld_dout, const float *dmean, float *dfeats_nhwc, void *workspace,
                            size_t workspace_bytes, void *stream)
{
    if (n_valid < 0 || batch < 1 || n_views < 1 || channels < 1 || !dfeats_nhwc || !krcam || !origin) return EPRECON_ERR_ARG;
    if (mode == EPRECON_BP_VARIANCE && !feats_nhwc) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t elems = (size_t)n_views * batch * height * width * channels;
    if (workspace && workspace_bytes < elems * sizeof(unsigned long long)) return EPRECON_ERR_WORKSPACE;
    if (workspace) EP_HIP_CHECK(hipMemsetAsync(workspace, 0, elems * sizeof(unsigned long long), st));
    // (the fp32 output is zeroed in the fixed-point form too: it carries the NaN marks of non-finite contributions, fixed_add)
    EP_HIP_CHECK(hipMemsetAsync(dfeats_nhwc, 0, elems * sizeof(float), st));
    if (n_valid == 0) return EPRECON_OK;
    if (!coords_valid || !dout) return EPRECON_ERR_ARG;
    BpBwdParams p;
    p.coords = coords_valid; p.n = n_valid; p.origin = origin; p.batch = batch; p.voxel_size = voxel_size;
    p.feats_nhwc = feats_nhwc; p.krcam = krcam; p.V = n_views; p.C = channels; p.H = height; p.W = width; p.mode = mode;
    p.dout = dout; p.ld_dout = ld_dout; p.dmean = dmean; p.dfeats = dfeats_nhwc; p.dfix = (unsigned long long *)workspace;
    const size_t lds = (((size_t)n_views * batch * 12 + 3) & ~(size_t)3) * sizeof(float);
    hipLaunchKernelGGL(bp_backward_kernel, dim3((unsigned)ceil_div(n_valid * channels, 256)), dim3(256), lds, st, p);
    EP_LAUNCH_CHECK();
    if (workspace) {
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)ceil_div((int64_t)elems, 256)), dim3(256), 0, st,
                           (const unsigned long long *)workspace, (long long)elems, dfeats_nhwc);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}

int eprecon_back_project_backward_async(const int32_t *coords_valid, int64_t n_valid, const float *origin, int batch,
                                        float voxel_size, const float *feats_nhwc, const float *krcam, int n_views,
                                        int channels, int height, int width, int mode, const float *dout, int ld_dout,
                                        const float *dmean, float *dfeats_nhwc, void *stream)
{
    return bp_backward_impl(coords_valid, n_valid, origin, batch, voxel_size, feats_nhwc, krcam, n_views, channels, height, width, mode,
                            dout, ld_dout, dmean, dfeats_nhwc, nullptr, 0, stream);
}

size_t eprecon_back_project_backward_workspace_bytes(int batch, int n_views, int channels, int height, int width)
{
    return (size_t)n_views * batch * height * width * channels * sizeof(unsigned long long);
}

int eprecon_back_project_backward_det_async(const int32_t *coords_valid, int64_t n_valid, const float *origin, int batch,
                                            float voxel_size, const float *feats_nhwc, const float *krcam, int n_views,
                                            int channels, int height, int width, int mode, const float *dout, int ld_dout,
                                            const float *dmean, float *dfeats_nhwc, void *workspace, size_t workspace_bytes,
                                            void *stream)
{
    if (!workspace) return EPRECON_ERR_ARG;
    return bp_backward_impl(coords_valid, n_valid, origin, batch, voxel_size, feats_nhwc, krcam, n_views, channels, height, width, mode,
                            dout, ld_dout, dmean, dfeats_nhwc, workspace, workspace_bytes, stream);
}

}  // extern "C"
