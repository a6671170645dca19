// Back-projection onto a DENSE voxel grid with LDS image patches per brick (gfx950).
//
// Same operator as back_project.hip (Back_Project.forward, models/occupancy_initialization.py:189-261), for the
// case in which the voxel list is the dense x-major raster of a (Dx, Dy, Dz) grid — what the reference's occupancy
// initialisation always back-projects (models/neucon_network.py:246-255, ops/generate_grids.py:3-10) and what
// BASELINE.json's configs[1] measures ("96^3 coarse back_project").  The coordinates are implicit, so the 16 bytes
// per voxel of the list are not read, and — the point of this file — the raster structure gives texel reuse:
//
//   bp_gather_mlp_kernel (back_project.hip) issues, per (voxel, view), four 96-byte gathers that each touch two
//   64-byte L1 lines: 72 L1 line accesses per voxel at C = 24, and the kernel runs at the L1 line rate.
//   Here a workgroup owns a BRICK of BX x 8 x 8 voxels.  Neighbouring voxels project ~1 texel apart, so the brick's
//   footprint in one view is a patch of ~12 x 12 texels: the patch is loaded ONCE into LDS with coalesced 16-byte
//   loads (~0.4 L1 line accesses per voxel and view instead of 8) and the 4 taps of every voxel are read from LDS.
//   Views are processed one after the other (patch + per-voxel tap records, one barrier pair per view); the running
//   sums stay in registers.  A view whose patch does not fit (grazing view, brick straddling the camera plane) falls
//   back to direct global gathers for that view.  Same fp32 arithmetic, same view order, same visible-view decisions
//   as the list kernels: the results are bit-identical (tests/test_back_project_gpu.py).
//
// Output order is the reference's: raster order of the valid voxels (stable compaction).  bp_dense_count_kernel writes
// the visible-view count of every voxel, per-256-voxel tile totals (scanned by bp_dense_scan_kernel) and the rank of
// every voxel inside its tile (one byte), from which a brick addresses its output rows.
#include "common.hpp"

namespace {
using namespace ep;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct DenseParams {
    int Dx, Dy, Dz, interval, batch;
    const float *origin;
    float voxel_size;
    const float *feats_nhwc;
    const float *krcam;
    int V, C, H, W, Cs, min_view;
    float *out_feats;
    int32_t *out_coords;
    float *count;
    int32_t *n_valid_dev;
    int32_t *tile_offsets;  // [ceil(N / 256)] exclusive scan of the valid totals per tile
    uint8_t *rank8;         // [N] rank of a voxel among the valid voxels of its 256-voxel tile
    int patch_texels;       // LDS patch capacity
};

struct ProjFast {
    float u, v, pz, uu, vu;  // u, v clamped into the image; uu, vu unclamped
    bool vis;
};

// identical arithmetic to back_project.hip:project_fast (bit-exact visibility, see there)
__device__ __forceinline__ ProjFast project_fast(const float *P, float X, float Y, float Z, float wm1, float hm1, float kx,
                                                 float ky)
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
    o.uu = u;
    o.vu = v;
    o.u = fminf(fmaxf(u, 0.0f), wm1);
    o.v = fminf(fmaxf(v, 0.0f), hm1);
    return o;
}

__device__ __forceinline__ void centre(const DenseParams &p, int b, int ix, int iy, int iz, float &X, float &Y, float &Z)
{
    X = __fadd_rn(__fmul_rn((float)(ix * p.interval), p.voxel_size), p.origin[3 * b + 0]);
    Y = __fadd_rn(__fmul_rn((float)(iy * p.interval), p.voxel_size), p.origin[3 * b + 1]);
    Z = __fadd_rn(__fmul_rn((float)(iz * p.interval), p.voxel_size), p.origin[3 * b + 2]);
}

// one thread per voxel (raster index): visible-view count, tile totals, in-tile ranks, per-batch totals
__global__ __launch_bounds__(256) void bp_dense_count_kernel(DenseParams p, int n, int32_t *tile_sums, int32_t *blk_batch)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sP = reinterpret_cast<float *>(smem);
    int *sWave = reinterpret_cast<int *>(sP + p.V * p.batch * 12);
    const int tid = threadIdx.x;
    for (int i = tid; i < p.V * p.batch * 12; i += 256) sP[i] = p.krcam[(i / 12) * 16 + (i % 12)];
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    bool valid = false;
    if (i < n) {
        const int iz = i % p.Dz, iy = (i / p.Dz) % p.Dy, ix = (i / (p.Dz * p.Dy)) % p.Dx, b = i / (p.Dz * p.Dy * p.Dx);
        float X, Y, Z;
        centre(p, b, ix, iy, iz, X, Y, Z);
        const float wm1 = (float)(p.W - 1), hm1 = (float)(p.H - 1);
        const float kx = 2.0f / wm1, ky = 2.0f / hm1;
        int cnt = 0;
        for (int v = 0; v < p.V; ++v) cnt += project_fast(sP + (v * p.batch + b) * 12, X, Y, Z, wm1, hm1, kx, ky).vis ? 1 : 0;
        p.count[i] = (float)cnt;
        valid = cnt >= p.min_view;
    }
    const unsigned long long m = __ballot(valid);
    const int lane = tid & 63, wid = tid >> 6;
    if (lane == 0) sWave[wid] = __popcll(m);
    __syncthreads();
    int before = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
        before += w < wid ? sWave[w] : 0;
        tot += sWave[w];
    }
    if (i < n) p.rank8[i] = (uint8_t)(before + __popcll(m & ((1ull << lane) - 1ull)));
    if (tid == 0) {
        tile_sums[blockIdx.x] = tot;
        // a 256-voxel tile lies inside one batch element when Dx*Dy*Dz % 256 == 0 (checked by the launcher)
        if (blk_batch) blk_batch[blockIdx.x] = tot;
    }
}

// exclusive scan of the tile totals (one workgroup) + n_valid_dev[0] and the per-batch totals
__global__ __launch_bounds__(1024) void bp_dense_scan_kernel(int32_t *tile_sums, int ntile, int32_t *n_valid_dev, int batch,
                                                             int tiles_per_batch)
{
    __shared__ int sWave[16];
    __shared__ int sBatch[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid < 16) sBatch[tid] = 0;
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < ntile; base += 1024) {
        const int i = base + tid;
        const int v = i < ntile ? tile_sums[i] : 0;
        if (i < ntile && v && batch > 1) atomicAdd(&sBatch[min(i / tiles_per_batch, 15)], v);
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) sWave[wid] = x;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            woff += w < wid ? sWave[w] : 0;
            tot += sWave[w];
        }
        if (i < ntile) tile_sums[i] = carry + woff + x - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) n_valid_dev[0] = carry;
    if (batch == 1) {
        if (tid == 0) n_valid_dev[1] = carry;
    } else if (tid < batch) {
        n_valid_dev[1 + tid] = sBatch[tid];
    }
}

constexpr int kDirect = 0x40000000;   // tap record flag: byte offset into the maps (direct gather), else LDS float offset
constexpr int kHidden = (int)0x80000000u;

// Q = C / 4 (16-byte channel groups), VOXB = voxels per workgroup = BX * 8 * 8
template <int Q, int VOXB, bool VARIANCE>
__global__ __launch_bounds__(256) void bp_gather_brick_kernel(DenseParams p, float *out_mean)
{
    constexpr int BX = VOXB / 64;
    constexpr int IT = VOXB * Q / 256;  // (voxel, channel group) items per thread
    constexpr int C = Q * 4;
    constexpr int CP = C + 4;           // LDS texel pitch in floats (16-byte aligned, breaks the power-of-two stride)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sPatch = reinterpret_cast<float *>(smem);                       // [patch_texels][CP]
    float *sP = sPatch + (size_t)p.patch_texels * CP;                      // [V][12] of this batch element
    int *sOff = reinterpret_cast<int *>(sP + ((p.V * 12 + 3) & ~3));       // [VOXB] tap record: offset / flags
    float2 *sWxy = reinterpret_cast<float2 *>(sOff + VOXB);                // [VOXB] fractional weights
    int *sRow = reinterpret_cast<int *>(sWxy + VOXB);                      // [VOXB] output row or -1
    float *sDen = reinterpret_cast<float *>(sRow + VOXB);                  // [VOXB] max(visible views, 1)
    float *sCorner = sDen + VOXB;                                          // [V][8][3] u, v, pz of the brick corners
    int *sBox = reinterpret_cast<int *>(sCorner + p.V * 8 * 3);            // [V][5] patch origin, size, mode

    const int tid = threadIdx.x;
    const int nbx = p.Dx / BX, nby = p.Dy / 8, nbz = p.Dz / 8;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int bz = lb % nbz, by = (lb / nbz) % nby, bxk = (lb / (nbz * nby)) % nbx, b = lb / (nbz * nby * nbx);
    const int x0v = bxk * BX, y0v = by * 8, z0v = bz * 8;
    for (int i = tid; i < p.V * 12; i += 256) sP[i] = p.krcam[((i / 12) * p.batch + b) * 16 + (i % 12)];
    const float wm1 = (float)(p.W - 1), hm1 = (float)(p.H - 1);
    const float kx = 2.0f / wm1, ky = 2.0f / hm1;
    const int map_elems = p.H * p.W * p.Cs;

    // ---- prologue: output rows, coordinates, visible-view counts of the brick's voxels ----
    for (int l = tid; l < VOXB; l += 256) {
        const int lz = l & 7, ly = (l >> 3) & 7, lx = l >> 6;
        const int ix = x0v + lx, iy = y0v + ly, iz = z0v + lz;
        const int i = ((b * p.Dx + ix) * p.Dy + iy) * p.Dz + iz;
        const float cnt = p.count[i];
        const bool valid = cnt >= (float)p.min_view;
        int row = -1;
        if (valid) {
            row = p.tile_offsets[i >> 8] + (int)p.rank8[i];
            reinterpret_cast<int4 *>(p.out_coords)[row] = make_int4(b, ix * p.interval, iy * p.interval, iz * p.interval);
        }
        sRow[l] = row;
        sDen[l] = fmaxf(cnt, 1.0f);
    }
    // ---- footprints of the brick in all views up front: the 8 corner voxels bound every voxel of the (convex) brick ----
    __syncthreads();   // sP is staged
    for (int e = tid; e < p.V * 8; e += 256) {
        const int v = e >> 3, k = e & 7;
        float X, Y, Z;
        centre(p, b, x0v + ((k & 4) ? BX - 1 : 0), y0v + ((k & 2) ? 7 : 0), z0v + ((k & 1) ? 7 : 0), X, Y, Z);
        const ProjFast q = project_fast(sP + v * 12, X, Y, Z, wm1, hm1, kx, ky);
        sCorner[e * 3 + 0] = q.uu; sCorner[e * 3 + 1] = q.vu; sCorner[e * 3 + 2] = q.pz;
    }
    __syncthreads();
    if (tid < p.V) {
        float umin = 1e30f, umax = -1e30f, vmin = 1e30f, vmax = -1e30f, zmin = 1e30f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float *c3 = sCorner + (tid * 8 + k) * 3;
            umin = fminf(umin, c3[0]); umax = fmaxf(umax, c3[0]);
            vmin = fminf(vmin, c3[1]); vmax = fmaxf(vmax, c3[1]);
            zmin = fminf(zmin, c3[2]);
        }
        const bool front = zmin > 1e-3f && umin > -1e6f && umax < 1e6f && vmin > -1e6f && vmax < 1e6f;
        int px0 = 0, py0 = 0, bw = 0, bh = 0, mode = 0;   // mode 0: direct gathers, 1: LDS patch, 2: brick outside the image
        if (front) {
            // texel range of all taps (x0 .. x0 + 1) with one texel of slack for rounding, clipped to the image
            px0 = max((int)floorf(umin) - 1, 0);
            py0 = max((int)floorf(vmin) - 1, 0);
            bw = min((int)floorf(umax) + 2, p.W - 1) - px0 + 1;
            bh = min((int)floorf(vmax) + 2, p.H - 1) - py0 + 1;
            mode = (bw <= 0 || bh <= 0) ? 2 : (bw * bh <= p.patch_texels ? 1 : 0);
        }
        sBox[tid * 5 + 0] = px0; sBox[tid * 5 + 1] = py0; sBox[tid * 5 + 2] = bw; sBox[tid * 5 + 3] = bh; sBox[tid * 5 + 4] = mode;
    }
    __syncthreads();

    float4 acc[IT], acc2[VARIANCE ? IT : 1];
#pragma unroll
    for (int it = 0; it < IT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.feats_nhwc), 0, p.V * p.batch * map_elems * 4, 0x00020000);
    const int s10 = p.Cs * 4, s01 = p.W * p.Cs * 4, s11 = s01 + s10;
    constexpr int NPL = 6;  // 16-byte patch elements per thread: patch_texels * Q <= 1536 by construction

    // the patch of view v -> registers (global loads in flight while the previous view is accumulated) -> LDS
    auto next_view = [&](int v) {
        while (v < p.V && sBox[v * 5 + 4] == 2) ++v;   // views that do not see the brick at all
        return v;
    };
    auto load_patch = [&](int v, float4(&regs)[NPL]) {
        if (v >= p.V || sBox[v * 5 + 4] != 1) return;
        const int px0 = sBox[v * 5], py0 = sBox[v * 5 + 1], bw = sBox[v * 5 + 2], ntex = bw * sBox[v * 5 + 3];
        const float *map = p.feats_nhwc + (size_t)(v * p.batch + b) * map_elems;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int e = min(tid + k * 256, ntex * Q - 1);   // clamped: a fixed number of loads in flight
            const int t = e / Q, q = e - t * Q;
            const int ty = t / bw, tx = t - ty * bw;
            regs[k] = *reinterpret_cast<const float4 *>(map + (size_t)((py0 + ty) * p.W + px0 + tx) * p.Cs + q * 4);
        }
    };
    auto store_patch = [&](int v, const float4(&regs)[NPL]) {
        if (sBox[v * 5 + 4] != 1) return;
        const int ntex = sBox[v * 5 + 2] * sBox[v * 5 + 3];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int e = tid + k * 256;
            if (e < ntex * Q) {
                const int t = e / Q, q = e - t * Q;
                *reinterpret_cast<float4 *>(sPatch + t * CP + q * 4) = regs[k];
            }
        }
    };

    // VARIANCE: pass 0 accumulates the sum (-> mean), pass 1 the squared deviations (two-sweep, like the reference)
    for (int pass = 0; pass < (VARIANCE ? 2 : 1); ++pass) {
        if (VARIANCE && pass == 1) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const float den = sDen[(tid + it * 256) / Q];
                acc[it] = make_float4(__fdiv_rn(acc[it].x, den), __fdiv_rn(acc[it].y, den), __fdiv_rn(acc[it].z, den),
                                      __fdiv_rn(acc[it].w, den));
                acc2[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float4 regs[NPL];
        int v = next_view(0);
        load_patch(v, regs);
        while (v < p.V) {
            const float *P = sP + v * 12;
            const int px0 = sBox[v * 5], py0 = sBox[v * 5 + 1], bw = sBox[v * 5 + 2], bh = sBox[v * 5 + 3];
            const bool use_patch = sBox[v * 5 + 4] == 1;
            const int px1 = px0 + bw - 1, py1 = py0 + bh - 1;
            const int map_base = (v * p.batch + b) * map_elems;
            __syncthreads();  // the previous view's patch and tap records are consumed
            store_patch(v, regs);
            // ---- tap records of the brick's voxels for this view ----
            for (int l = tid; l < VOXB; l += 256) {
                const int lz = l & 7, ly = (l >> 3) & 7, lx = l >> 6;
                int off = kHidden;
                if (sRow[l] >= 0) {
                    float X, Y, Z;
                    centre(p, b, x0v + lx, y0v + ly, z0v + lz, X, Y, Z);
                    const ProjFast q = project_fast(P, X, Y, Z, wm1, hm1, kx, ky);
                    if (q.vis) {
                        const float x0f = fminf(floorf(q.u), wm1 - 1.0f), y0f = fminf(floorf(q.v), hm1 - 1.0f);
                        sWxy[l] = make_float2(q.u - x0f, q.v - y0f);
                        const int xi = (int)x0f, yi = (int)y0f;
                        const bool inpatch = use_patch && xi >= px0 && xi + 1 <= px1 && yi >= py0 && yi + 1 <= py1;
                        // direct offsets are in floats, tagged with the flag bit (the maps hold < 2^30 floats: launcher check)
                        off = inpatch ? ((yi - py0) * bw + (xi - px0)) * CP : (kDirect | (map_base + (yi * p.W + xi) * p.Cs));
                    }
                }
                sOff[l] = off;
            }
            __syncthreads();
            const int vn = next_view(v + 1);
            load_patch(vn, regs);   // in flight during the accumulation below
            // ---- accumulate: item (voxel, 4 channels) ----
            const int rowp = bw * CP;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int item = tid + it * 256;
                const int vox = item / Q, q = item - vox * Q;
                const int off = sOff[vox];
                if (off == kHidden) continue;
                float4 a, bb, c, d;
                if (off & kDirect) {
                    const int byte = ((off & ~kDirect) + q * 4) * 4;
                    const u32x4 ra = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte, 0, 0);
                    const u32x4 rb = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte, s10, 0);
                    const u32x4 rc = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte, s01, 0);
                    const u32x4 rd = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte, s11, 0);
                    a = make_float4(__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(ra.z), __uint_as_float(ra.w));
                    bb = make_float4(__uint_as_float(rb.x), __uint_as_float(rb.y), __uint_as_float(rb.z), __uint_as_float(rb.w));
                    c = make_float4(__uint_as_float(rc.x), __uint_as_float(rc.y), __uint_as_float(rc.z), __uint_as_float(rc.w));
                    d = make_float4(__uint_as_float(rd.x), __uint_as_float(rd.y), __uint_as_float(rd.z), __uint_as_float(rd.w));
                } else {
                    const float *t = sPatch + off + q * 4;
                    a = *reinterpret_cast<const float4 *>(t);
                    bb = *reinterpret_cast<const float4 *>(t + CP);
                    c = *reinterpret_cast<const float4 *>(t + rowp);
                    d = *reinterpret_cast<const float4 *>(t + rowp + CP);
                }
                const float2 w = sWxy[vox];
                const float wx1 = w.x, wy1 = w.y, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
                const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
                float4 r;
                r.x = fmaf(d.x, w11, fmaf(c.x, w01, fmaf(bb.x, w10, a.x * w00)));
                r.y = fmaf(d.y, w11, fmaf(c.y, w01, fmaf(bb.y, w10, a.y * w00)));
                r.z = fmaf(d.z, w11, fmaf(c.z, w01, fmaf(bb.z, w10, a.z * w00)));
                r.w = fmaf(d.w, w11, fmaf(c.w, w01, fmaf(bb.w, w10, a.w * w00)));
                if (VARIANCE && pass == 1) {
                    const float dx = r.x - acc[it].x, dy = r.y - acc[it].y, dz = r.z - acc[it].z, dw = r.w - acc[it].w;
                    acc2[it].x = fmaf(dx, dx, acc2[it].x); acc2[it].y = fmaf(dy, dy, acc2[it].y);
                    acc2[it].z = fmaf(dz, dz, acc2[it].z); acc2[it].w = fmaf(dw, dw, acc2[it].w);
                } else {
                    acc[it].x += r.x; acc[it].y += r.y; acc[it].z += r.z; acc[it].w += r.w;
                }
            }
            v = vn;
        }
        __syncthreads();   // the last view's records are consumed before the next pass rewrites them
    }
    // ---- epilogue: mean (or variance + mean) rows ----
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int item = tid + it * 256;
        const int vox = item / Q, q = item - vox * Q;
        const int row = sRow[vox];
        if (row < 0) continue;
        const float den = sDen[vox];
        if (VARIANCE) {
            *reinterpret_cast<float4 *>(p.out_feats + (size_t)row * C + q * 4) =
                make_float4(__fdiv_rn(acc2[it].x, den), __fdiv_rn(acc2[it].y, den), __fdiv_rn(acc2[it].z, den),
                            __fdiv_rn(acc2[it].w, den));
            if (out_mean) *reinterpret_cast<float4 *>(out_mean + (size_t)row * C + q * 4) = acc[it];
        } else {
            *reinterpret_cast<float4 *>(p.out_feats + (size_t)row * C + q * 4) =
                make_float4(__fdiv_rn(acc[it].x, den), __fdiv_rn(acc[it].y, den), __fdiv_rn(acc[it].z, den),
                            __fdiv_rn(acc[it].w, den));
        }
    }
}

template <int Q, int VOXB>
int launch_brick(const DenseParams &p, int mode, float *out_mean, hipStream_t st)
{
    constexpr int BX = VOXB / 64;
    const int nbrick = p.batch * (p.Dx / BX) * (p.Dy / 8) * (p.Dz / 8);
    const size_t lds = (size_t)p.patch_texels * (Q * 4 + 4) * 4 + (size_t)((p.V * 12 + 3) & ~3) * 4 + (size_t)VOXB * (4 + 8 + 4 + 4) +
                       (size_t)p.V * (8 * 3 + 5) * 4 + 16;
    if (mode == EPRECON_BP_VARIANCE)
        hipLaunchKernelGGL((bp_gather_brick_kernel<Q, VOXB, true>), dim3(nbrick), dim3(256), lds, st, p, out_mean);
    else
        hipLaunchKernelGGL((bp_gather_brick_kernel<Q, VOXB, false>), dim3(nbrick), dim3(256), lds, st, p, out_mean);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // namespace

extern "C" size_t eprecon_back_project_dense_workspace_bytes(int64_t n, int batch)
{
    return ep::align_up((size_t)ep::ceil_div(n > 0 ? n : 1, 256) * 4, 256) * 2 + ep::align_up((size_t)(n > 0 ? n : 1), 256) + 256;
}

extern "C" int eprecon_back_project_dense_async(const int32_t *dims_host, int interval, const float *origin, int batch,
                                                float voxel_size, const float *feats_nhwc, const float *krcam, int n_views,
                                                int channels, int height, int width, int min_view, int mode,
                                                float *out_feats, float *out_mean, int32_t *out_coords, float *count,
                                                int32_t *n_valid_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!dims_host || !origin || !feats_nhwc || !krcam || !out_feats || !out_coords || !count || !n_valid_dev || !workspace ||
        interval <= 0 || batch <= 0 || batch > 15 || n_views <= 0 || n_views > 32 || height <= 1 || width <= 1)
        return EPRECON_ERR_ARG;
    if (mode != EPRECON_BP_MEAN && mode != EPRECON_BP_VARIANCE) return EPRECON_ERR_UNSUPPORTED;
    const int Dx = dims_host[0], Dy = dims_host[1], Dz = dims_host[2];
    const int64_t per_batch = (int64_t)Dx * Dy * Dz, n = per_batch * batch;
    const int Q = channels / 4;
    // brick shape by channel count: (voxels x channel groups) / 256 accumulators per thread stay <= 16
    const int voxb = Q <= 8 ? 512 : (Q <= 16 ? 256 : 128);
    if (channels % 4 || (Q != 6 && Q != 8 && Q != 10 && Q != 20) || Dy % 8 || Dz % 8 || Dx % (voxb / 64) || per_batch % 256 ||
        n > 0x7fffffff || (size_t)n_views * batch * channels * height * width >= (1ull << 30))
        return EPRECON_ERR_UNSUPPORTED;
    if (workspace_bytes < eprecon_back_project_dense_workspace_bytes(n, batch)) return EPRECON_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    const int ntile = (int)(n / 256);
    int32_t *tile_sums = reinterpret_cast<int32_t *>(ws);
    ws += 2 * ep::align_up((size_t)ntile * 4, 256);
    DenseParams p;
    p.Dx = Dx; p.Dy = Dy; p.Dz = Dz; p.interval = interval; p.batch = batch; p.origin = origin; p.voxel_size = voxel_size;
    p.feats_nhwc = feats_nhwc; p.krcam = krcam; p.V = n_views; p.C = channels; p.H = height; p.W = width; p.Cs = channels;
    p.min_view = min_view; p.out_feats = out_feats; p.out_coords = out_coords; p.count = count; p.n_valid_dev = n_valid_dev;
    p.tile_offsets = tile_sums; p.rank8 = reinterpret_cast<uint8_t *>(ws);
    p.patch_texels = (24 * 1024) / ((channels + 4) * 4);
    const size_t lds_count = ((size_t)n_views * batch * 12 + 8) * 4;
    hipLaunchKernelGGL(bp_dense_count_kernel, dim3(ntile), dim3(256), lds_count, st, p, (int)n, tile_sums, (int32_t *)nullptr);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bp_dense_scan_kernel, dim3(1), dim3(1024), 0, st, tile_sums, ntile, n_valid_dev, batch,
                       (int)(per_batch / 256));
    EP_LAUNCH_CHECK();
    int rc = ep::profile_bracket_begin(st);   // bench.py's roofline hook (back_project.hip)
    if (rc != EPRECON_OK) return rc;
    switch (Q) {
        case 6: rc = launch_brick<6, 512>(p, mode, out_mean, st); break;
        case 8: rc = launch_brick<8, 512>(p, mode, out_mean, st); break;
        case 10: rc = launch_brick<10, 256>(p, mode, out_mean, st); break;
        default: rc = launch_brick<20, 128>(p, mode, out_mean, st); break;
    }
    if (rc != EPRECON_OK) return rc;
    return ep::profile_bracket_end(st, "bp_gather_brick_kernel");
}
