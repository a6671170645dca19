// Point <-> voxel transfers of the point-voxel U-Net (SPVCNN) and the sparse ConvGRU on gfx950.
//
// Replaces (reference call sites; kernels are in the un-vendored torchsparse, semantics restated
// in SURVEY.md appendix A.2 and DESIGN.md):
//   aligned-camera point coordinates                 models/neucon_network.py:387-398, models/gru_fusion.py:332-337
//   initial_voxelize  (floor, unique, scatter-mean)   ops/torchsparse_utils.py:15-35
//   point_to_voxel    (hash query + scatter-mean)     ops/torchsparse_utils.py:40-63
//   voxel_to_point    (8-corner query, trilinear weights, weighted gather)   ops/torchsparse_utils.py:68-105
//
// Scatter-mean is done without floating-point atomics: a CSR list of the points of every voxel is
// built (integer atomics only, then each short list is sorted by point index), and one thread per
// (voxel, 4 channels) sums its points in index order -> deterministic, and the same lists serve
// every later point_to_voxel on that voxel set.  All kernels are bandwidth/latency bound.
#include "hashgrid.hpp"

namespace {
using namespace ep;

__device__ __forceinline__ int floor_div_i(int a, int q) { return (a >= 0) ? a / q : -((-a + q - 1) / q); }

// r[0:3] = W[:3,:] . [c * vs + origin, 1] with the k-ordered fma chain of a [N,4]@[4,3] matmul
__global__ void aligned_coords_kernel(const int4 *coords, int n, const float *origin, float vs,
                                      const float *w2ac, int batch, float4 *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = coords[i];
    const int b = min(max(c.x, 0), batch - 1);
    const float X = __fadd_rn(__fmul_rn((float)c.y, vs), origin[3 * b + 0]);
    const float Y = __fadd_rn(__fmul_rn((float)c.z, vs), origin[3 * b + 1]);
    const float Z = __fadd_rn(__fmul_rn((float)c.w, vs), origin[3 * b + 2]);
    const float *M = w2ac + 16 * b;
    float r[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
        r[j] = __fmaf_rn(1.0f, M[4 * j + 3], __fmaf_rn(Z, M[4 * j + 2], __fmaf_rn(Y, M[4 * j + 1], __fmul_rn(X, M[4 * j]))));
    out[i] = make_float4(r[0], r[1], r[2], (float)c.x);
}

// pts f32[N,4] (x,y,z,b) -> scaled f32[N,4] (x/res, y/res, z/res, b) and int32[N,4] (b, floor ...)
// (n_dev, optional: the live count is min(n, *n_dev); the launch is sized by n)
__global__ void point_quantize_kernel(const float4 *pts, int n, float res, float4 *scaled, int4 *vox, const int32_t *n_dev)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (i >= n) return;
    const float4 p = pts[i];
    const float x = __fdiv_rn(p.x, res), y = __fdiv_rn(p.y, res), z = __fdiv_rn(p.z, res);
    scaled[i] = make_float4(x, y, z, p.w);
    vox[i] = make_int4((int)p.w, (int)floorf(x), (int)floorf(y), (int)floorf(z));
}

// The coordinate side of an SPVCNN pass from a voxel list whose LENGTH is still on the device (the rows a compaction has just
// written: sparsify's kept rows, a back-projection's valid rows): [8 children per row, models/neucon_network.py:193-214 ->]
// aligned-camera points (:387-398) -> scaled points and voxel indices (ops/torchsparse_utils.py:15-19) in ONE launch sized by
// the capacity; thread 0 leaves the live point count (8 x rows or rows) for the dn numbering calls that follow.  The arithmetic
// is that of upsample_coords_kernel (grid_ops.hip), aligned_coords_kernel and point_quantize_kernel, statement by statement.
__global__ void spvcnn_points_dn_kernel(const int4 *src, int cap_src, const int32_t *n_src_dev, int children, int interval,
                                        const float *origin, float vs, const float *w2ac, int batch, float res, int4 *up,
                                        float4 *r_out, float4 *scaled, int4 *vox, int32_t *n_pts_dev)
{
    const int n_src = min(cap_src, max(*n_src_dev, 0));
    const int n_pts = children ? n_src * 8 : n_src;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) n_pts_dev[0] = n_pts;
    if (e >= n_pts) return;
    int4 c;
    if (children) {
        const int i = e >> 3, k = e & 7;
        const int dx = (0xB2 >> k) & 1, dy = (0xD4 >> k) & 1, dz = (0xE8 >> k) & 1;   // child order 0, x, y, z, xy, xz, yz, xyz
        c = src[i];
        c.y += dx * interval;
        c.z += dy * interval;
        c.w += dz * interval;
        up[e] = c;
    } else {
        c = src[e];
    }
    const int b = min(max(c.x, 0), batch - 1);
    const float X = __fadd_rn(__fmul_rn((float)c.y, vs), origin[3 * b + 0]);
    const float Y = __fadd_rn(__fmul_rn((float)c.z, vs), origin[3 * b + 1]);
    const float Z = __fadd_rn(__fmul_rn((float)c.w, vs), origin[3 * b + 2]);
    const float *M = w2ac + 16 * b;
    float r[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
        r[j] = __fmaf_rn(1.0f, M[4 * j + 3], __fmaf_rn(Z, M[4 * j + 2], __fmaf_rn(Y, M[4 * j + 1], __fmul_rn(X, M[4 * j]))));
    const float pw = (float)c.x;
    r_out[e] = make_float4(r[0], r[1], r[2], pw);
    const float x = __fdiv_rn(r[0], res), y = __fdiv_rn(r[1], res), z = __fdiv_rn(r[2], res);
    scaled[e] = make_float4(x, y, z, pw);
    vox[e] = make_int4((int)pw, (int)floorf(x), (int)floorf(y), (int)floorf(z));
}

// ---- CSR point lists per voxel ----
__global__ void seg_count_kernel(const int32_t *idx, int n, int32_t *counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] >= 0) atomicAdd(&counts[idx[i]], 1);
}
__global__ void seg_fill_kernel(const int32_t *idx, int n, const int32_t *offsets, int32_t *cursor,
                                int32_t *order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = idx[i];
    if (v < 0) return;
    order[offsets[v] + atomicAdd(&cursor[v], 1)] = i;
}
// Rank sort of every voxel's point list, one thread per POINT: its slot is the number of points of the same
// voxel with a smaller index.  (An insertion sort by one thread per voxel cost 60 us on average and far more at
// the coarse strides, where a voxel holds up to 512 points: O(L^2) sequential steps.  Here the same
// comparisons are spread over the L threads of the list, which only read.)
__global__ void seg_rank_kernel(const int32_t *idx, int n, const int32_t *offsets, const int32_t *unsorted,
                                int32_t *order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = idx[i];
    if (v < 0) return;
    const int a = offsets[v], b = offsets[v + 1];
    int rank = 0;
    for (int t = a; t < b; ++t) rank += unsorted[t] < i ? 1 : 0;
    order[a + rank] = i;
}

// out[v, c] = mean over the points of voxel v (index order) of feat[p, c]
__global__ __launch_bounds__(256) void seg_mean_kernel(const float *feat, int ld_f, const int32_t *offsets,
                                                       const int32_t *order, int m, int C, float *out,
                                                       int ld_o)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)m * C) return;
    const int v = (int)(e / C), c = (int)(e - (int64_t)v * C);
    const int a = offsets[v], b = offsets[v + 1];
    float s = 0.0f;
    for (int i = a; i < b; ++i) s += feat[(size_t)order[i] * ld_f + c];
    out[(size_t)v * ld_o + c] = (b > a) ? __fdiv_rn(s, (float)(b - a)) : 0.0f;
}

// the same with one thread per (voxel, 4 channels): 16-byte loads / stores, the list indices are read once per 4 channels
__global__ __launch_bounds__(256) void seg_mean4_kernel(const float *feat, int ld_f, const int32_t *offsets,
                                                        const int32_t *order, int m, int C4, float *out, int ld_o)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)m * C4) return;
    const int v = (int)(e / C4), c = 4 * (int)(e - (int64_t)v * C4);
    const int a = offsets[v], b = offsets[v + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = a; i < b; ++i) {
        const float4 f = *reinterpret_cast<const float4 *>(feat + (size_t)order[i] * ld_f + c);
        s.x += f.x; s.y += f.y; s.z += f.z; s.w += f.w;
    }
    const float n = (float)(b - a);
    if (b > a) s = make_float4(__fdiv_rn(s.x, n), __fdiv_rn(s.y, n), __fdiv_rn(s.z, n), __fdiv_rn(s.w, n));
    *reinterpret_cast<float4 *>(out + (size_t)v * ld_o + c) = s;
}

// 8-corner lookup + trilinear weights at tensor stride s:
//   base = floor(p / s) * s, corner k = base + (bx, by, bz) * s with k = 4 bx + 2 by + bz,
//   w_k = prod over axes of (bit ? p - pf : pc - p), / s^3 when s != 1, 0 for absent corners,
//   then w /= (sum w + 1e-8).
__global__ void trilinear_map_kernel(HashTable t, const float4 *pts, int n, int stride, int32_t *idx,
                                     float *wts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const float s = (float)stride;
    const float xf = floorf(__fdiv_rn(p.x, s)) * s, yf = floorf(__fdiv_rn(p.y, s)) * s,
                zf = floorf(__fdiv_rn(p.z, s)) * s;
    const float xc = xf + s, yc = yf + s, zc = zf + s;
    const int bx = (int)xf, by = (int)yf, bz = (int)zf, b = (int)p.w;
    float w[8];
    int id[8];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ox = (k >> 2) & 1, oy = (k >> 1) & 1, oz = k & 1;
        const int x = bx + ox * stride, y = by + oy * stride, z = bz + oz * stride;
        id[k] = key_in_range(b, x, y, z) ? hash_lookup(t, pack_key(b, x, y, z)) : -1;
        float wk = (ox ? (p.x - xf) : (xc - p.x)) * (oy ? (p.y - yf) : (yc - p.y)) * (oz ? (p.z - zf) : (zc - p.z));
        if (stride != 1) wk = __fdiv_rn(wk, s * s * s);
        if (id[k] < 0) wk = 0.0f;
        w[k] = wk;
        sum += wk;
    }
    const float den = sum + 1e-8f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        idx[(size_t)i * 8 + k] = id[k];
        wts[(size_t)i * 8 + k] = __fdiv_rn(w[k], den);
    }
}

// The same table WITHOUT hash probes, for points whose base voxel row is already known and whose voxel set has its 3x3x3 kernel
// map built: corner (ox, oy, oz) of the base voxel is the map's neighbour at offset (+ox, +oy, +oz) — nbr[((oz + 1) 3 + (oy + 1)) 3
// + ox + 1][base] — eight reads of a table that is about to be read by every convolution on the set anyway, instead of eight
// probes of the hash grid.  base: the point's own voxel at stride 1 (the unique numbering's inverse), the result of the stride-s
// hash query of its voxel otherwise.  Same indices, same weights, bit for bit (trilinear_map_kernel's arithmetic).
__global__ void trilinear_from_map_kernel(const float4 *pts, int n, const int32_t *base, const int32_t *nbr, int m, int stride,
                                          int32_t *idx, float *wts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const float s = (float)stride;
    const float xf = floorf(__fdiv_rn(p.x, s)) * s, yf = floorf(__fdiv_rn(p.y, s)) * s,
                zf = floorf(__fdiv_rn(p.z, s)) * s;
    const float xc = xf + s, yc = yf + s, zc = zf + s;
    const int bi = base[i];
    float w[8];
    int id[8];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ox = (k >> 2) & 1, oy = (k >> 1) & 1, oz = k & 1;
        const int kk = ((oz + 1) * 3 + (oy + 1)) * 3 + (ox + 1);
        id[k] = (bi >= 0 && bi < m) ? nbr[(size_t)kk * m + bi] : -1;
        float wk = (ox ? (p.x - xf) : (xc - p.x)) * (oy ? (p.y - yf) : (yc - p.y)) * (oz ? (p.z - zf) : (zc - p.z));
        if (stride != 1) wk = __fdiv_rn(wk, s * s * s);
        if (id[k] < 0) wk = 0.0f;
        w[k] = wk;
        sum += wk;
    }
    const float den = sum + 1e-8f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        idx[(size_t)i * 8 + k] = id[k];
        wts[(size_t)i * 8 + k] = __fdiv_rn(w[k], den);
    }
}

// out[i, c] (+)= sum_k w[i,k] * feat[idx[i,k], c]
__global__ __launch_bounds__(256) void devoxelize_kernel(const float *feat, int ld_f, const int32_t *idx,
                                                         const float *wts, int n, int C, float *out,
                                                         int ld_o, int accumulate)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * C) return;
    const int i = (int)(e / C), c = (int)(e - (int64_t)i * C);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = idx[(size_t)i * 8 + k];
        if (j >= 0) s = fmaf(wts[(size_t)i * 8 + k], feat[(size_t)j * ld_f + c], s);
    }
    float *o = out + (size_t)i * ld_o + c;
    *o = accumulate ? (*o + s) : s;
}

// SConv3d's tail inside ConvGRU (models/modules.py:193-196,214-221): v = devoxelise(conv) + Linear(z.F) (the Linear
// output arrives in `skip`), followed by the gate arithmetic the reference does with separate tensor ops:
//   mode 1  out = sigmoid(v)                         update gate z
//   mode 2  out = sigmoid(v) * h                     r * h, written straight into the [r*h, x] concat buffer
//   mode 3  out = (1 - zg) * h + zg * tanh(v)        new hidden state
// tail (optional): tail_dst[i, 0:tail_c] = tail_src[i, 0:tail_c] in the same launch — the x half of the [r*h, x] buffer the
// reference builds with torch.cat (models/modules.py:218), so that no separate copy of the [h, x] buffer is needed
__global__ __launch_bounds__(256) void devoxelize_gate_kernel(const float *feat, int ld_f, const int32_t *idx,
                                                              const float *wts, int n, int C, const float *skip,
                                                              int ld_s, int mode, const float *h, int ld_h,
                                                              const float *zg, int ld_z, float *out, int ld_o,
                                                              const float *tail_src, int ld_ts, float *tail_dst, int ld_td, int tail_c)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * C) return;
    const int i = (int)(e / C), c = (int)(e - (int64_t)i * C);
    for (int t = c; t < tail_c; t += C) tail_dst[(size_t)i * ld_td + t] = tail_src[(size_t)i * ld_ts + t];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = idx[(size_t)i * 8 + k];
        if (j >= 0) s = fmaf(wts[(size_t)i * 8 + k], feat[(size_t)j * ld_f + c], s);
    }
    const float v = s + skip[(size_t)i * ld_s + c];
    float r;
    if (mode == 3) {
        const float z = zg[(size_t)i * ld_z + c], hv = h[(size_t)i * ld_h + c];
        r = (1.0f - z) * hv + z * tanhf(v);
    } else {
        r = __fdiv_rn(1.0f, 1.0f + expf(-v));
        if (mode == 2) r *= h[(size_t)i * ld_h + c];
    }
    out[(size_t)i * ld_o + c] = r;
}

// 4 channels per thread (16-byte loads; the 8 indices / weights of a point are read once per 4 channels)
__device__ __forceinline__ float4 devox4(const float *feat, int ld_f, const int32_t *idx, const float *wts, int i, int c)
{
    const int4 i0 = *reinterpret_cast<const int4 *>(idx + (size_t)i * 8), i1 = *reinterpret_cast<const int4 *>(idx + (size_t)i * 8 + 4);
    const float4 w0 = *reinterpret_cast<const float4 *>(wts + (size_t)i * 8), w1 = *reinterpret_cast<const float4 *>(wts + (size_t)i * 8 + 4);
    const int jj[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (jj[k] >= 0) {
            const float4 f = *reinterpret_cast<const float4 *>(feat + (size_t)jj[k] * ld_f + c);
            s.x = fmaf(ww[k], f.x, s.x); s.y = fmaf(ww[k], f.y, s.y); s.z = fmaf(ww[k], f.z, s.z); s.w = fmaf(ww[k], f.w, s.w);
        }
    }
    return s;
}

__global__ __launch_bounds__(256) void devoxelize4_kernel(const float *feat, int ld_f, const int32_t *idx, const float *wts,
                                                          int n, int C4, float *out, int ld_o, int accumulate)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * C4) return;
    const int i = (int)(e / C4), c = 4 * (int)(e - (int64_t)i * C4);
    float4 s = devox4(feat, ld_f, idx, wts, i, c);
    float4 *o = reinterpret_cast<float4 *>(out + (size_t)i * ld_o + c);
    if (accumulate) {
        const float4 p = *o;
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *o = s;
}

__device__ __forceinline__ float gate1(float v, int mode, float h, float z)
{
    if (mode == 3) return (1.0f - z) * h + z * tanhf(v);
    const float r = __fdiv_rn(1.0f, 1.0f + expf(-v));
    return mode == 2 ? r * h : r;
}

__global__ __launch_bounds__(256) void devoxelize_gate4_kernel(const float *feat, int ld_f, const int32_t *idx,
                                                               const float *wts, int n, int C4, const float *skip, int ld_s,
                                                               int mode, const float *h, int ld_h, const float *zg, int ld_z,
                                                               float *out, int ld_o, const float *tail_src, int ld_ts,
                                                               float *tail_dst, int ld_td, int tail_c4)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * C4) return;
    const int i = (int)(e / C4), c = 4 * (int)(e - (int64_t)i * C4);
    for (int t = c / 4; t < tail_c4; t += C4)
        *reinterpret_cast<float4 *>(tail_dst + (size_t)i * ld_td + 4 * t) = *reinterpret_cast<const float4 *>(tail_src + (size_t)i * ld_ts + 4 * t);
    const float4 s = devox4(feat, ld_f, idx, wts, i, c);
    const float4 k = *reinterpret_cast<const float4 *>(skip + (size_t)i * ld_s + c);
    float4 hv = make_float4(0.f, 0.f, 0.f, 0.f), zv = hv;
    if (mode >= 2) hv = *reinterpret_cast<const float4 *>(h + (size_t)i * ld_h + c);
    if (mode == 3) zv = *reinterpret_cast<const float4 *>(zg + (size_t)i * ld_z + c);
    float4 r;
    r.x = gate1(s.x + k.x, mode, hv.x, zv.x);
    r.y = gate1(s.y + k.y, mode, hv.y, zv.y);
    r.z = gate1(s.z + k.z, mode, hv.z, zv.z);
    r.w = gate1(s.w + k.w, mode, hv.w, zv.w);
    *reinterpret_cast<float4 *>(out + (size_t)i * ld_o + c) = r;
}

__host__ inline bool vec4_ok(const void *p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; }

HashTable make_table(const void *mem, uint32_t cap)
{
    HashTable t;
    t.keys = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(const_cast<void *>(mem)) + 256);
    t.vals = reinterpret_cast<int32_t *>(t.keys + cap);
    t.mask = cap - 1;
    return t;
}

// torchsparse's coordinate hash (F.sphash, ops/torchsparse_utils.py:19): FNV-1a over the four int32 of a row
// in (x, y, z, batch) order, folded to 60 bits.  Needed only to reproduce the ORDER in which the
// reference numbers voxels (`torch.unique(pc_hash)` = ascending hash), see remap_stale_index_kernel.
__global__ void sphash_kernel(const int4 *coords_bxyz, int n, long long *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = coords_bxyz[i];
    const int v[4] = {c.y, c.z, c.w, c.x};
    unsigned long long h = 14695981039346656037ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h ^= (unsigned int)v[j];
        h *= 1099511628211ull;
    }
    h = (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFull);
    out[i] = (long long)h;
}

// ConvGRU's second gate convolution in the reference devoxelises with the corner indices cached by the
// FIRST voxelisation (ops/torchsparse_utils.py:70-71,97-99) into the SECOND voxel set's feature rows; both
// sets are numbered by ascending sphash.  With this build's first-occurrence numbering the same rows are
// reached through  old id -> hash rank in the old set -> id of the voxel with that rank in the new set.
// Ranks beyond the new set (out-of-bounds reads in the reference) become -1 (no contribution).
__global__ void remap_stale_index_kernel(const int32_t *idx, long long n, const int32_t *rank_old,
                                         const int32_t *perm_new, int m_new, int32_t *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx[i];
    int r = -1;
    if (j >= 0) {
        const int k = rank_old[j];
        r = k < m_new ? perm_new[k] : -1;
    }
    out[i] = r;
}

}  // namespace

extern "C" {

int eprecon_aligned_coords_async(const int32_t *coords, int64_t n, const float *origin, int batch,
                                 float voxel_size, const float *world_to_aligned_camera, float *out_xyzb,
                                 void *stream)
{
    if (n < 0 || batch <= 0 || !origin || !world_to_aligned_camera || (n > 0 && (!coords || !out_xyzb)))
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(aligned_coords_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const int4 *>(coords), (int)n, origin,
                       voxel_size, world_to_aligned_camera, batch, reinterpret_cast<float4 *>(out_xyzb));
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_point_quantize_async(const float *points_xyzb, int64_t n, float resolution, float *scaled_xyzb,
                                 int32_t *voxel_bxyz, void *stream)
{
    if (n < 0 || !(resolution > 0.0f) || (n > 0 && (!points_xyzb || !scaled_xyzb || !voxel_bxyz)))
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(point_quantize_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const float4 *>(points_xyzb), (int)n, resolution,
                       reinterpret_cast<float4 *>(scaled_xyzb), reinterpret_cast<int4 *>(voxel_bxyz), (const int32_t *)nullptr);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_point_quantize_dn_async(const float *points_xyzb, int64_t n_cap, const int32_t *n_dev, float resolution,
                                    float *scaled_xyzb, int32_t *voxel_bxyz, void *stream)
{
    if (n_cap < 0 || !n_dev || !(resolution > 0.0f) || (n_cap > 0 && (!points_xyzb || !scaled_xyzb || !voxel_bxyz)))
        return EPRECON_ERR_ARG;
    if (n_cap == 0) return EPRECON_OK;
    hipLaunchKernelGGL(point_quantize_kernel, dim3((unsigned)ceil_div(n_cap, 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const float4 *>(points_xyzb), (int)n_cap, resolution,
                       reinterpret_cast<float4 *>(scaled_xyzb), reinterpret_cast<int4 *>(voxel_bxyz), n_dev);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_spvcnn_points_dn_async(const int32_t *src_coords, int64_t cap_src, const int32_t *n_src_dev, int children,
                                   int interval, const float *origin, int batch, float voxel_size,
                                   const float *world_to_aligned_camera, float resolution, int32_t *up_coords, float *r_coords,
                                   float *scaled_xyzb, int32_t *voxel_bxyz, int32_t *n_points_dev, void *stream)
{
    if (cap_src < 0 || !n_src_dev || batch <= 0 || !origin || !world_to_aligned_camera || !(resolution > 0.0f) || !n_points_dev ||
        (children && (interval <= 0 || !up_coords)) || (cap_src > 0 && (!src_coords || !r_coords || !scaled_xyzb || !voxel_bxyz)))
        return EPRECON_ERR_ARG;
    const int64_t cap_pts = children ? cap_src * 8 : cap_src;
    if (cap_pts > 0x7fffffff) return EPRECON_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(spvcnn_points_dn_kernel, dim3((unsigned)ceil_div(cap_pts > 0 ? cap_pts : 1, 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const int4 *>(src_coords), (int)cap_src, n_src_dev, children ? 1 : 0,
                       interval, origin, voxel_size, world_to_aligned_camera, batch, resolution, reinterpret_cast<int4 *>(up_coords),
                       reinterpret_cast<float4 *>(r_coords), reinterpret_cast<float4 *>(scaled_xyzb),
                       reinterpret_cast<int4 *>(voxel_bxyz), n_points_dev);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

size_t eprecon_segment_workspace_bytes(int64_t n, int64_t m)
{
    return align_up((size_t)(m > 0 ? m : 1) * 4, 256) * 2 + align_up((size_t)ceil_div(m > 0 ? m : 1, 2048) * 4, 256) +
           align_up((size_t)(n > 0 ? n : 1) * 4, 256) + 256;
}

/* CSR lists: offsets int32[m+1], order int32[n] (first offsets[m] entries used) */
int eprecon_segment_lists_async(const int32_t *idx, int64_t n, int64_t m, int32_t *offsets, int32_t *order,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || m < 0 || !offsets || !workspace || (n > 0 && (!idx || !order))) return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_segment_workspace_bytes(n, m)) return EPRECON_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        EP_HIP_CHECK(hipMemsetAsync(offsets, 0, sizeof(int32_t), st));
        return EPRECON_OK;
    }
    char *ws = reinterpret_cast<char *>(workspace);
    const size_t seg = align_up((size_t)m * 4, 256);
    int32_t *counts = reinterpret_cast<int32_t *>(ws);
    int32_t *cursor = reinterpret_cast<int32_t *>(ws + seg);
    int32_t *scratch = reinterpret_cast<int32_t *>(ws + 2 * seg);
    int32_t *unsorted = reinterpret_cast<int32_t *>(ws + 2 * seg + align_up((size_t)ceil_div(m, 2048) * 4, 256));
    EP_HIP_CHECK(hipMemsetAsync(counts, 0, 2 * seg, st));
    const dim3 gn((unsigned)ceil_div(n > 0 ? n : 1, 256)), blk(256);
    if (n > 0) {
        hipLaunchKernelGGL(seg_count_kernel, gn, blk, 0, st, idx, (int)n, counts);
        EP_LAUNCH_CHECK();
    }
    int rc = ep::exclusive_scan_i32(counts, (int)m, offsets, scratch, offsets + m, st);
    if (rc != EPRECON_OK) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(seg_fill_kernel, gn, blk, 0, st, idx, (int)n, (const int32_t *)offsets, cursor, unsorted);
        EP_LAUNCH_CHECK();
        // offsets[m] (grand total) was written by the scan
        hipLaunchKernelGGL(seg_rank_kernel, gn, blk, 0, st, idx, (int)n, (const int32_t *)offsets,
                           (const int32_t *)unsorted, order);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}

int eprecon_segment_mean_async(const float *feat, int ld_feat, const int32_t *offsets, const int32_t *order,
                               int64_t m, int channels, float *out, int ld_out, void *stream)
{
    if (m < 0 || channels <= 0 || !offsets || (m > 0 && (!feat || !order || !out)) || ld_feat < channels ||
        ld_out < channels)
        return EPRECON_ERR_ARG;
    if (m == 0) return EPRECON_OK;
    if ((channels & 3) == 0 && vec4_ok(feat, ld_feat) && vec4_ok(out, ld_out))
        hipLaunchKernelGGL(seg_mean4_kernel, dim3((unsigned)ceil_div(m * (channels / 4), 256)), dim3(256), 0,
                           (hipStream_t)stream, feat, ld_feat, offsets, order, (int)m, channels / 4, out, ld_out);
    else
        hipLaunchKernelGGL(seg_mean_kernel, dim3((unsigned)ceil_div(m * channels, 256)), dim3(256), 0,
                           (hipStream_t)stream, feat, ld_feat, offsets, order, (int)m, channels, out, ld_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_trilinear_map_async(const void *table, uint32_t capacity, const float *points_xyzb, int64_t n,
                                int stride, int32_t *idx8, float *weight8, void *stream)
{
    if (!table || capacity < 1024 || (capacity & (capacity - 1)) || n < 0 || stride < 1 ||
        (n > 0 && (!points_xyzb || !idx8 || !weight8)))
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(trilinear_map_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, make_table(table, capacity),
                       reinterpret_cast<const float4 *>(points_xyzb), (int)n, stride, idx8, weight8);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"

namespace ep {
int trilinear_from_map(const float *points_xyzb, int64_t n, const int32_t *base_row, const int32_t *nbr27, int64_t m, int stride,
                       int32_t *idx8, float *weight8, void *stream)
{
    if (n < 0 || m < 0 || m > 0x7fffffff || stride < 1 || (n > 0 && (!points_xyzb || !base_row || !nbr27 || !idx8 || !weight8)))
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(trilinear_from_map_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(points_xyzb), (int)n, base_row, nbr27, (int)m, stride, idx8, weight8);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}
}  // namespace ep

extern "C" {

int eprecon_devoxelize_async(const float *voxel_feat, int ld_feat, const int32_t *idx8, const float *weight8,
                             int64_t n, int channels, float *out, int ld_out, int accumulate, void *stream)
{
    if (n < 0 || channels <= 0 || (n > 0 && (!voxel_feat || !idx8 || !weight8 || !out)) ||
        ld_feat < channels || ld_out < channels)
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    if ((channels & 3) == 0 && vec4_ok(voxel_feat, ld_feat) && vec4_ok(out, ld_out) && vec4_ok(idx8, 4) && vec4_ok(weight8, 4))
        hipLaunchKernelGGL(devoxelize4_kernel, dim3((unsigned)ceil_div(n * (channels / 4), 256)), dim3(256), 0,
                           (hipStream_t)stream, voxel_feat, ld_feat, idx8, weight8, (int)n, channels / 4, out, ld_out, accumulate);
    else
        hipLaunchKernelGGL(devoxelize_kernel, dim3((unsigned)ceil_div(n * channels, 256)), dim3(256), 0,
                           (hipStream_t)stream, voxel_feat, ld_feat, idx8, weight8, (int)n, channels, out, ld_out,
                           accumulate);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_sphash_async(const int32_t *coords, int64_t n, int64_t *out_hash, void *stream)
{
    if (n < 0 || (n > 0 && (!coords || !out_hash))) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(sphash_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const int4 *>(coords), (int)n, reinterpret_cast<long long *>(out_hash));
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_remap_index_async(const int32_t *idx, int64_t n, const int32_t *rank_old, const int32_t *perm_new,
                              int64_t m_new, int32_t *out, void *stream)
{
    if (n < 0 || m_new < 0 || (n > 0 && (!idx || !rank_old || !out || (m_new > 0 && !perm_new)))) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(remap_stale_index_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       idx, (long long)n, rank_old, perm_new, (int)m_new, out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}


static int devoxelize_gate_impl(const float *voxel_feat, int ld_feat, const int32_t *idx8, const float *weight8,
                                int64_t n, int channels, const float *skip, int ld_skip, int mode, const float *h,
                                int ld_h, const float *zgate, int ld_z, float *out, int ld_out, const float *tail_src,
                                int ld_tail_src, float *tail_dst, int ld_tail_dst, int tail_channels, void *stream)
{
    if (n < 0 || channels <= 0 || mode < 1 || mode > 3 || tail_channels < 0) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    if (!voxel_feat || !idx8 || !weight8 || !skip || !out || (mode >= 2 && !h) || (mode == 3 && !zgate))
        return EPRECON_ERR_ARG;
    if (tail_channels > 0 && (!tail_src || !tail_dst || ld_tail_src < tail_channels || ld_tail_dst < tail_channels)) return EPRECON_ERR_ARG;
    const bool v4 = (channels & 3) == 0 && vec4_ok(voxel_feat, ld_feat) && vec4_ok(out, ld_out) && vec4_ok(skip, ld_skip) &&
                    vec4_ok(idx8, 4) && vec4_ok(weight8, 4) && (mode < 2 || vec4_ok(h, ld_h)) && (mode < 3 || vec4_ok(zgate, ld_z)) &&
                    (tail_channels == 0 || ((tail_channels & 3) == 0 && vec4_ok(tail_src, ld_tail_src) && vec4_ok(tail_dst, ld_tail_dst)));
    if (v4)
        hipLaunchKernelGGL(devoxelize_gate4_kernel, dim3((unsigned)ceil_div(n * (channels / 4), 256)), dim3(256), 0,
                           (hipStream_t)stream, voxel_feat, ld_feat, idx8, weight8, (int)n, channels / 4, skip, ld_skip, mode,
                           h, ld_h, zgate, ld_z, out, ld_out, tail_src, ld_tail_src, tail_dst, ld_tail_dst, tail_channels / 4);
    else
        hipLaunchKernelGGL(devoxelize_gate_kernel, dim3((unsigned)ceil_div(n * channels, 256)), dim3(256), 0,
                           (hipStream_t)stream, voxel_feat, ld_feat, idx8, weight8, (int)n, channels, skip, ld_skip, mode,
                           h, ld_h, zgate, ld_z, out, ld_out, tail_src, ld_tail_src, tail_dst, ld_tail_dst, tail_channels);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_devoxelize_gate_async(const float *voxel_feat, int ld_feat, const int32_t *idx8, const float *weight8,
                                  int64_t n, int channels, const float *skip, int ld_skip, int mode, const float *h,
                                  int ld_h, const float *zgate, int ld_z, float *out, int ld_out, void *stream)
{
    return devoxelize_gate_impl(voxel_feat, ld_feat, idx8, weight8, n, channels, skip, ld_skip, mode, h, ld_h, zgate, ld_z, out,
                                ld_out, nullptr, 0, nullptr, 0, 0, stream);
}

int eprecon_devoxelize_gate_tail_async(const float *voxel_feat, int ld_feat, const int32_t *idx8, const float *weight8,
                                       int64_t n, int channels, const float *skip, int ld_skip, int mode, const float *h,
                                       int ld_h, const float *zgate, int ld_z, float *out, int ld_out, const float *tail_src,
                                       int ld_tail_src, float *tail_dst, int ld_tail_dst, int tail_channels, void *stream)
{
    return devoxelize_gate_impl(voxel_feat, ld_feat, idx8, weight8, n, channels, skip, ld_skip, mode, h, ld_h, zgate, ld_z, out,
                                ld_out, tail_src, ld_tail_src, tail_dst, ld_tail_dst, tail_channels, stream);
}

}  // extern "C"
