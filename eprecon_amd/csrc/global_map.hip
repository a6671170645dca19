// Persistent sparse global map of GRUFusion behind an opaque handle (gfx950).
//
// Replaces the map state and its per-fragment bookkeeping in the reference (models/gru_fusion.py):
//   self.global_volume[scale] = PointTensor(F [M,C], C [M,3])              :31-38, reset :59-65
//   convert2dense: crop to the fragment bounding volume (FBV), dense scatter of map and fragment,
//                  nonzero union, gathers                                    :67-114, :321-326
//   update_map:   map = map[outside the FBV] ++ (union coords + relative origin, fused rows)   :195-215
//   the ground-truth twin (target_tsdf_volume, default 1, stored where |tsdf| < 1)             :99-113, :206-213
// The reference re-creates the map tensors with torch.cat every fragment; here the rows live in two
// ping-pong device buffers owned by the handle (geometric growth), the crop marks the rows inside
// the FBV, and the update is a stable compaction of the rows outside followed by the appended rows —
// the SAME row order as the reference's `cat([old[valid == False], new])`, which parity tests pin.
// No dense feature volume is ever built (only int32 index volumes of the FBV, as in fbv_union.hip).
#include <new>

#include "common.hpp"

namespace {
using namespace ep;

struct EpMap {
    int channels = 0;
    int64_t size = 0, cap = 0;
    int32_t *coords[2] = {nullptr, nullptr};  // [cap,3] scene-grid voxel units of this scale
    float *feats[2] = {nullptr, nullptr};     // [cap,channels]
    int cur = 0;
    int32_t *keep = nullptr, *keep_rank = nullptr;  // [row_cap] 1 = row outside the last crop's FBV; its scan
    int64_t row_cap = 0;
    int32_t *scan_scratch = nullptr;
    int64_t scratch_cap = 0;
    int32_t *counts_dev = nullptr;  // [4]
    int32_t *counts_host = nullptr;  // pinned [4]
    // dense FBV workspace: idx_cur, idx_glob, flag, rank (int32 [cells] each) + vol (f32 [cells])
    char *dense = nullptr;
    size_t dense_bytes = 0;
    int64_t kept = -1;  // rows outside the FBV of the last crop (-1: no crop pending, -2: pending, count still on the device)
    int pending_dim = 0;  // grid size of the pending ground-truth crop (eprecon_gru_stage_begin_async -> _commit_async)
    int rel[3] = {0, 0, 0};
    // multi-GPU boundary exchange: per row, which fragment produced its features and whether THIS rank
    // fused it: 0 unknown, +(fragment + 1) fused here, -(fragment + 1) received from another rank
    int32_t *stamps[2] = {nullptr, nullptr};  // [cap], allocated with the rows
    int fuse_stamp = 0;                       // what eprecon_map_update_async writes for the rows it appends
    int32_t *sel = nullptr, *sel_rank = nullptr, *sel_aux = nullptr;  // [sel_cap] selection flags of the exchange, their scan, merge scratch
    int64_t sel_cap = 0;
    int32_t *sel_scratch = nullptr;
    int64_t sel_scratch_cap = 0;
    int64_t n_selected = -1;
};

int ensure_rows(EpMap *m, int64_t rows)
{
    if (rows <= m->cap) return EPRECON_OK;
    int64_t cap = m->cap > 0 ? m->cap : 4096;
    while (cap < rows) cap *= 2;
    for (int b = 0; b < 2; ++b) {
        int32_t *c = nullptr, *s = nullptr;
        float *f = nullptr;
        EP_HIP_CHECK(hipMalloc(&c, (size_t)cap * 3 * sizeof(int32_t)));
        EP_HIP_CHECK(hipMalloc(&f, (size_t)cap * m->channels * sizeof(float)));
        EP_HIP_CHECK(hipMalloc(&s, (size_t)cap * sizeof(int32_t)));
        EP_HIP_CHECK(hipMemset(s, 0, (size_t)cap * sizeof(int32_t)));
        if (b == m->cur && m->size > 0) {  // only the live buffer carries data
            EP_HIP_CHECK(hipMemcpy(c, m->coords[b], (size_t)m->size * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice));
            EP_HIP_CHECK(hipMemcpy(f, m->feats[b], (size_t)m->size * m->channels * sizeof(float),
                                   hipMemcpyDeviceToDevice));
            EP_HIP_CHECK(hipMemcpy(s, m->stamps[b], (size_t)m->size * sizeof(int32_t), hipMemcpyDeviceToDevice));
        }
        if (m->coords[b]) EP_HIP_CHECK(hipFree(m->coords[b]));
        if (m->feats[b]) EP_HIP_CHECK(hipFree(m->feats[b]));
        if (m->stamps[b]) EP_HIP_CHECK(hipFree(m->stamps[b]));
        m->coords[b] = c;
        m->feats[b] = f;
        m->stamps[b] = s;
    }
    m->cap = cap;
    return EPRECON_OK;
}

int ensure_flags(EpMap *m, int64_t rows, int64_t scan_n)
{
    if (rows > m->row_cap) {
        int64_t cap = m->row_cap > 0 ? m->row_cap : 4096;
        while (cap < rows) cap *= 2;
        if (m->keep) EP_HIP_CHECK(hipFree(m->keep));
        if (m->keep_rank) EP_HIP_CHECK(hipFree(m->keep_rank));
        EP_HIP_CHECK(hipMalloc(&m->keep, (size_t)cap * sizeof(int32_t)));
        EP_HIP_CHECK(hipMalloc(&m->keep_rank, (size_t)cap * sizeof(int32_t)));
        m->row_cap = cap;
    }
    const int64_t need = ceil_div(scan_n > rows ? scan_n : rows, 2048) + 8;
    if (need > m->scratch_cap) {
        if (m->scan_scratch) EP_HIP_CHECK(hipFree(m->scan_scratch));
        EP_HIP_CHECK(hipMalloc(&m->scan_scratch, (size_t)need * 2 * sizeof(int32_t)));
        m->scratch_cap = need * 2;
    }
    return EPRECON_OK;
}

int ensure_dense(EpMap *m, int dim)
{
    const size_t cells = (size_t)dim * dim * dim;
    const size_t need = 5 * align_up(cells * 4, 256);
    if (need > m->dense_bytes) {
        if (m->dense) EP_HIP_CHECK(hipFree(m->dense));
        EP_HIP_CHECK(hipMalloc(&m->dense, need));
        m->dense_bytes = need;
    }
    return EPRECON_OK;
}

__device__ __forceinline__ bool row_active(const float *row, int C, int g, int mode)
{
    bool nz = false;
    for (int c = g; c < C; c += 8) nz |= mode ? (fabsf(row[c]) < 1.0f) : (row[c] != 0.0f);
    unsigned long long msk = __ballot(nz);
    const int lane = threadIdx.x & 63;
    return ((msk >> (lane & ~7)) & 0xFFull) != 0ull;
}

struct CropParams {
    const int32_t *cur_coords;
    const float *cur_feat;
    int n_cur, ld_cur;
    const int32_t *glob_coords;
    const float *glob_feat;
    int n_glob, C, D, interval, mode;
    int rel[3];
    int32_t *idx_cur, *idx_glob, *flag, *keep;
};

// 8 lanes per row; the same activity rule and cell addressing as fbv_union.hip, plus keep[j] for the
// map rows (1 = outside the FBV = survives update_map untouched)
__global__ __launch_bounds__(256) void map_crop_scatter_kernel(CropParams p)
{
    const int g = threadIdx.x & 7;
    const int r = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int total = p.n_cur + p.n_glob;
    const bool live = r < total;
    const bool is_cur = r < p.n_cur;
    const int j = is_cur ? r : r - p.n_cur;
    int x = -1, y = -1, z = -1;
    const float *row = p.cur_feat;
    if (live) {
        if (is_cur) {
            x = p.cur_coords[4 * j + 1] / p.interval;
            y = p.cur_coords[4 * j + 2] / p.interval;
            z = p.cur_coords[4 * j + 3] / p.interval;
            row = p.cur_feat + (size_t)j * p.ld_cur;
        } else {
            x = p.glob_coords[3 * j + 0] - p.rel[0];
            y = p.glob_coords[3 * j + 1] - p.rel[1];
            z = p.glob_coords[3 * j + 2] - p.rel[2];
            row = p.glob_feat + (size_t)j * p.C;
        }
    }
    const bool inside = live && x >= 0 && x < p.D && y >= 0 && y < p.D && z >= 0 && z < p.D;
    const bool nz = row_active(row, (live ? p.C : 0), g, p.mode);
    if (live && g == 0) {
        if (!is_cur) p.keep[j] = inside ? 0 : 1;
        if (inside) {
            const int cell = (x * p.D + y) * p.D + z;
            (is_cur ? p.idx_cur : p.idx_glob)[cell] = j;
            if (nz) p.flag[cell] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void map_emit_kernel(const int32_t *flag, const int32_t *rank, const int32_t *idx_cur,
                                                       const int32_t *idx_glob, int D, int32_t *updated,
                                                       int32_t *src_cur, int32_t *src_glob)
{
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= D * D * D || !flag[cell]) return;
    const int o = rank[cell];
    updated[3 * o + 0] = cell / (D * D);
    updated[3 * o + 1] = (cell / D) % D;
    updated[3 * o + 2] = cell % D;
    src_cur[o] = idx_cur[cell];
    src_glob[o] = idx_glob[cell];
}

__global__ __launch_bounds__(256) void map_gather_kernel(const float *feat, int ld_f, int col0, const int32_t *src, int n,
                                                         int C, float fill, float *out, int ld_o)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * C) return;
    const int i = (int)(e / C), c = (int)(e - (int64_t)i * C);
    const int j = src[i];
    out[(size_t)i * ld_o + c] = j >= 0 ? feat[(size_t)j * ld_f + col0 + c] : fill;
}

// stable compaction of the kept rows (old order) into the other buffer
__global__ __launch_bounds__(256) void map_compact_kernel(const int32_t *keep, const int32_t *keep_rank, int n,
                                                          const int32_t *c_in, const float *f_in, const int32_t *s_in, int C,
                                                          int32_t *c_out, float *f_out, int32_t *s_out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int W = C + 3;
    if (e >= (int64_t)n * W) return;
    const int j = (int)(e / W), c = (int)(e - (int64_t)j * W);
    if (!keep[j]) return;
    const int o = keep_rank[j];
    if (c < 3) {
        c_out[3 * (size_t)o + c] = c_in[3 * (size_t)j + c];
        if (c == 0) s_out[o] = s_in[j];
    } else {
        f_out[(size_t)o * C + (c - 3)] = f_in[(size_t)j * C + (c - 3)];
    }
}

__global__ __launch_bounds__(256) void map_append_kernel(const int32_t *updated, const float *values, int ld_v, int n, int C,
                                                         int rx, int ry, int rz, int64_t base, int stamp, int32_t *c_out,
                                                         float *f_out, int32_t *s_out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int W = C + 3;
    if (e >= (int64_t)n * W) return;
    const int i = (int)(e / W), c = (int)(e - (int64_t)i * W);
    const size_t o = (size_t)(base + i);
    if (c < 3) {
        c_out[3 * o + c] = updated[3 * (size_t)i + c] + (c == 0 ? rx : (c == 1 ? ry : rz));
        if (c == 0) s_out[o] = stamp;
    } else {
        f_out[o * C + (c - 3)] = values[(size_t)i * ld_v + (c - 3)];
    }
}

// ---- boundary exchange: selection, packing, merge (replace the boolean-indexing / sort / searchsorted glue) ----
// flag the rows this rank fused itself that lie inside any OTHER rank's fragment bounding volume
__global__ __launch_bounds__(256) void map_select_kernel(const int32_t *coords, const int32_t *stamps, int n,
                                                         const int32_t *boxes, int nbox, int skip_box, int D, int32_t *sel)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int hit = 0;
    if (stamps[j] > 0) {
        const int x = coords[3 * j], y = coords[3 * j + 1], z = coords[3 * j + 2];
        for (int b = 0; b < nbox && !hit; ++b) {
            if (b == skip_box) continue;
            const int lx = boxes[3 * b], ly = boxes[3 * b + 1], lz = boxes[3 * b + 2];
            hit = x >= lx && x < lx + D && y >= ly && y < ly + D && z >= lz && z < lz + D;
        }
    }
    sel[j] = hit;
}
// payload row = (x, y, z, fragment index: int32 bit patterns | C feature floats), selected rows in map order
__global__ __launch_bounds__(256) void map_pack_kernel(const int32_t *sel, const int32_t *sel_rank, int n, const int32_t *coords,
                                                       const float *feats, const int32_t *stamps, int C, float *payload)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int W = C + 4;
    if (e >= (int64_t)n * W) return;
    const int j = (int)(e / W), c = (int)(e - (int64_t)j * W);
    if (!sel[j]) return;
    float *row = payload + (size_t)sel_rank[j] * W;
    if (c < 3) row[c] = __int_as_float(coords[3 * (size_t)j + c]);
    else if (c == 3) row[3] = __int_as_float(stamps[j] - 1);
    else row[c] = feats[(size_t)j * C + (c - 4)];
}
__global__ void map_index_kernel(const int32_t *coords, int n, int D, int lx, int ly, int lz, int32_t *idx)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int x = coords[3 * j] - lx, y = coords[3 * j + 1] - ly, z = coords[3 * j + 2] - lz;
    if (x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D) idx[(x * D + y) * D + z] = j;
}
__device__ __forceinline__ int payload_cell(const float *row, int D, int lx, int ly, int lz)
{
    const int x = __float_as_int(row[0]) - lx, y = __float_as_int(row[1]) - ly, z = __float_as_int(row[2]) - lz;
    return (x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D) ? (x * D + y) * D + z : -1;
}
// newest received copy per cell: best[cell] = max(fragment + 1)
__global__ void merge_best_kernel(const float *payload, int n, int W, int D, int lx, int ly, int lz, int32_t *best)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *row = payload + (size_t)i * W;
    const int cell = payload_cell(row, D, lx, ly, lz);
    if (cell >= 0) atomicMax(best + cell, __float_as_int(row[3]) + 1);
}
// the (unique) newest copy of a cell: overwrites the local row when it is newer, or is flagged for appending
__global__ void merge_claim_kernel(const float *payload, int n, int W, int D, int lx, int ly, int lz, int32_t *best,
                                   const int32_t *idx, const int32_t *stamps, int32_t *action)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *row = payload + (size_t)i * W;
    const int cell = payload_cell(row, D, lx, ly, lz);
    int act = 0;  // 0 drop, 1 append, 2 + row: overwrite that row
    if (cell >= 0) {
        const int v = __float_as_int(row[3]) + 1;
        if (atomicCAS(best + cell, v, -v) == v) {  // first claimant of the newest stamp
            const int j = idx[cell];
            if (j < 0) act = 1;
            else if (v > abs(stamps[j])) act = 2 + j;
        }
    }
    action[i] = act;
}
__global__ __launch_bounds__(256) void merge_apply_kernel(const float *payload, int n, int C, const int32_t *action,
                                                          const int32_t *add_rank, int64_t base, int32_t *coords, float *feats,
                                                          int32_t *stamps)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int W = C + 4;
    if (e >= (int64_t)n * W) return;
    const int i = (int)(e / W), c = (int)(e - (int64_t)i * W);
    const int act = action[i];
    if (act == 0) return;
    const float *row = payload + (size_t)i * W;
    const size_t o = act == 1 ? (size_t)(base + add_rank[i]) : (size_t)(act - 2);
    if (c < 3) {
        if (act == 1) coords[3 * o + c] = __float_as_int(row[c]);
    } else if (c == 3) {
        stamps[o] = -(__float_as_int(row[3]) + 1);  // received, not to be re-broadcast
    } else {
        feats[o * C + (c - 4)] = row[c];
    }
}
__global__ void action_to_flag_kernel(const int32_t *action, int n, int32_t *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = action[i] == 1;
}
__global__ void fill_i32_kernel(int32_t *p, int64_t n, int32_t v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int ensure_sel(EpMap *m, int64_t rows)
{
    if (rows > m->sel_cap) {
        int64_t cap = m->sel_cap > 0 ? m->sel_cap : 4096;
        while (cap < rows) cap *= 2;
        if (m->sel) EP_HIP_CHECK(hipFree(m->sel));
        if (m->sel_rank) EP_HIP_CHECK(hipFree(m->sel_rank));
        if (m->sel_aux) EP_HIP_CHECK(hipFree(m->sel_aux));
        EP_HIP_CHECK(hipMalloc(&m->sel, (size_t)cap * sizeof(int32_t)));
        EP_HIP_CHECK(hipMalloc(&m->sel_rank, (size_t)cap * sizeof(int32_t)));
        EP_HIP_CHECK(hipMalloc(&m->sel_aux, (size_t)cap * sizeof(int32_t)));
        m->sel_cap = cap;
    }
    const int64_t need = ceil_div(rows, 2048) + 8;
    if (need > m->sel_scratch_cap) {
        if (m->sel_scratch) EP_HIP_CHECK(hipFree(m->sel_scratch));
        EP_HIP_CHECK(hipMalloc(&m->sel_scratch, (size_t)need * sizeof(int32_t)));
        m->sel_scratch_cap = need;
    }
    return EPRECON_OK;
}

// ---- ground-truth twin ----
__global__ void fill_f32_kernel(float *p, int n, float v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// map rows inside the FBV -> dense volume; keep flags for the rows outside
__global__ void target_scatter_kernel(const int32_t *coords, const float *feat, int n, int D, int rx, int ry, int rz,
                                      float *vol, int32_t *keep)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int x = coords[3 * j] - rx, y = coords[3 * j + 1] - ry, z = coords[3 * j + 2] - rz;
    const bool inside = x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D;
    keep[j] = inside ? 0 : 1;
    if (inside) vol[(x * D + y) * D + z] = feat[j];
}
// the current fragment's ground truth overwrites the map's; flag = |v| < 1 (what update_map stores)
__global__ void target_merge_kernel(const float *tsdf_gt, const uint8_t *occ_gt, int cells, float *vol, int32_t *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    float v = vol[i];
    if (occ_gt[i]) {
        v = tsdf_gt[i];
        vol[i] = v;
    }
    flag[i] = fabsf(v) < 1.0f ? 1 : 0;
}
__global__ void target_lookup_kernel(const float *vol, const int32_t *updated, int n, int D, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = vol[(updated[3 * i] * D + updated[3 * i + 1]) * D + updated[3 * i + 2]];
}
__global__ void target_append_kernel(const float *vol, const int32_t *flag, const int32_t *rank, int D, int rx, int ry,
                                     int rz, int64_t base, int32_t *c_out, float *f_out, int32_t *s_out)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= D * D * D || !flag[cell]) return;
    const size_t o = (size_t)(base + rank[cell]);
    c_out[3 * o + 0] = cell / (D * D) + rx;
    c_out[3 * o + 1] = (cell / D) % D + ry;
    c_out[3 * o + 2] = cell % D + rz;
    f_out[o] = vol[cell];
    s_out[o] = 0;  // the ground-truth twin takes no part in the boundary exchange: origin unknown, never a stale stamp
}

// ---- GRU-fusion stage with device-side counts (eprecon_gru_stage_begin_async) ----
// [h | x] buffers of the two ConvGRUs of a scale in one pass: h = the map's row, x = the fragment's row (zeros where absent),
// channels [0, chv) to the voxel cell, [chv, C) to the image cell.  The union size lives on the device.
__global__ __launch_bounds__(256) void stage_gather_kernel(const float *map_feat, const float *cur_feat, int ld_cur,
                                                           const int32_t *src_glob, const int32_t *src_cur, int n_cap,
                                                           const int32_t *n_dev, int C, int chv, float *hx_v, float *hx_i)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n = min(n_cap, *n_dev);
    if (e >= (int64_t)n * C) return;
    const int i = (int)(e / C), c = (int)(e - (int64_t)i * C);
    const int jg = src_glob[i], jc = src_cur[i];
    const float h = jg >= 0 ? map_feat[(size_t)jg * C + c] : 0.0f;
    const float x = jc >= 0 ? cur_feat[(size_t)jc * ld_cur + c] : 0.0f;
    const int chi = C - chv;
    if (c < chv) {
        hx_v[(size_t)i * 2 * chv + c] = h;
        hx_v[(size_t)i * 2 * chv + chv + c] = x;
    } else {
        hx_i[(size_t)i * 2 * chi + (c - chv)] = h;
        hx_i[(size_t)i * 2 * chi + chi + (c - chv)] = x;
    }
}
// union cells -> voxel coordinates of the fragment (batch, cell * interval) and their aligned-camera coordinates
// (models/gru_fusion.py:332-337; the arithmetic of aligned_coords_kernel, csrc/voxelize.hip: separate multiply / add, then the
// k-ordered fma chain of the [N,4] x [4,3] product); the batch column of the points is 0 like the reference's
__global__ void stage_points_kernel(const int32_t *updated, int n_cap, const int32_t *n_dev, int interval, int batch_index,
                                    const float *origin, float vs, const float *w2ac, int4 *out_coords, float4 *r_coords)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(n_cap, *n_dev)) return;
    const int cx = updated[3 * i] * interval, cy = updated[3 * i + 1] * interval, cz = updated[3 * i + 2] * interval;
    out_coords[i] = make_int4(batch_index, cx, cy, cz);
    const float X = __fadd_rn(__fmul_rn((float)cx, vs), origin[0]);
    const float Y = __fadd_rn(__fmul_rn((float)cy, vs), origin[1]);
    const float Z = __fadd_rn(__fmul_rn((float)cz, vs), origin[2]);
    float r[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
        r[j] = __fmaf_rn(1.0f, w2ac[4 * j + 3], __fmaf_rn(Z, w2ac[4 * j + 2], __fmaf_rn(Y, w2ac[4 * j + 1], __fmul_rn(X, w2ac[4 * j]))));
    r_coords[i] = make_float4(r[0], r[1], r[2], 0.0f);
}
// ... and, in the same launch, the two voxelisations' coordinate side (point_quantize_kernel twice, csrc/voxelize.hip: IEEE division
// by the resolution, floor; the second one on the ALREADY-SCALED points — ConvGRU's convr, models/modules.py:216-217)
__global__ void stage_points_quantize_kernel(const int32_t *updated, int n_cap, const int32_t *n_dev, int interval, int batch_index,
                                             const float *origin, float vs, const float *w2ac, float res, int4 *out_coords,
                                             float4 *r_coords, float4 *scaled1, int4 *vox1, float4 *scaled2, int4 *vox2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(n_cap, *n_dev)) return;
    const int cx = updated[3 * i] * interval, cy = updated[3 * i + 1] * interval, cz = updated[3 * i + 2] * interval;
    out_coords[i] = make_int4(batch_index, cx, cy, cz);
    const float X = __fadd_rn(__fmul_rn((float)cx, vs), origin[0]);
    const float Y = __fadd_rn(__fmul_rn((float)cy, vs), origin[1]);
    const float Z = __fadd_rn(__fmul_rn((float)cz, vs), origin[2]);
    float r[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
        r[j] = __fmaf_rn(1.0f, w2ac[4 * j + 3], __fmaf_rn(Z, w2ac[4 * j + 2], __fmaf_rn(Y, w2ac[4 * j + 1], __fmul_rn(X, w2ac[4 * j]))));
    r_coords[i] = make_float4(r[0], r[1], r[2], 0.0f);
    const float x1 = __fdiv_rn(r[0], res), y1 = __fdiv_rn(r[1], res), z1 = __fdiv_rn(r[2], res);
    scaled1[i] = make_float4(x1, y1, z1, 0.0f);
    vox1[i] = make_int4(0, (int)floorf(x1), (int)floorf(y1), (int)floorf(z1));
    const float x2 = __fdiv_rn(x1, res), y2 = __fdiv_rn(y1, res), z2 = __fdiv_rn(z1, res);
    scaled2[i] = make_float4(x2, y2, z2, 0.0f);
    vox2[i] = make_int4(0, (int)floorf(x2), (int)floorf(y2), (int)floorf(z2));
}
__global__ void target_lookup_dn_kernel(const float *vol, const int32_t *updated, int n_cap, const int32_t *n_dev, int D, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(n_cap, *n_dev)) return;
    out[i] = vol[(updated[3 * i] * D + updated[3 * i + 1]) * D + updated[3 * i + 2]];
}

inline EpMap *as_map(void *h) { return reinterpret_cast<EpMap *>(h); }

}  // namespace

extern "C" {

int eprecon_map_create(int channels, void **out_handle)
{
    if (channels <= 0 || !out_handle) return EPRECON_ERR_ARG;
    EpMap *m = new (std::nothrow) EpMap();
    if (!m) return EPRECON_ERR_ARG;
    m->channels = channels;
    EP_HIP_CHECK(hipMalloc(&m->counts_dev, 4 * sizeof(int32_t)));
    EP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&m->counts_host), 4 * sizeof(int32_t), hipHostMallocDefault));
    *out_handle = m;
    return EPRECON_OK;
}

int eprecon_map_destroy(void *handle)
{
    EpMap *m = as_map(handle);
    if (!m) return EPRECON_OK;
    for (int b = 0; b < 2; ++b) {
        if (m->coords[b]) (void)hipFree(m->coords[b]);
        if (m->feats[b]) (void)hipFree(m->feats[b]);
        if (m->stamps[b]) (void)hipFree(m->stamps[b]);
    }
    if (m->sel) (void)hipFree(m->sel);
    if (m->sel_rank) (void)hipFree(m->sel_rank);
    if (m->sel_aux) (void)hipFree(m->sel_aux);
    if (m->sel_scratch) (void)hipFree(m->sel_scratch);
    if (m->keep) (void)hipFree(m->keep);
    if (m->keep_rank) (void)hipFree(m->keep_rank);
    if (m->scan_scratch) (void)hipFree(m->scan_scratch);
    if (m->counts_dev) (void)hipFree(m->counts_dev);
    if (m->counts_host) (void)hipHostFree(m->counts_host);
    if (m->dense) (void)hipFree(m->dense);
    delete m;
    return EPRECON_OK;
}

int eprecon_map_reset(void *handle)
{
    EpMap *m = as_map(handle);
    if (!m) return EPRECON_ERR_ARG;
    m->size = 0;
    m->kept = -1;
    return EPRECON_OK;
}

int64_t eprecon_map_size(const void *handle) { return handle ? reinterpret_cast<const EpMap *>(handle)->size : -1; }
int eprecon_map_channels(const void *handle) { return handle ? reinterpret_cast<const EpMap *>(handle)->channels : -1; }

int eprecon_map_export_async(const void *handle, int32_t *coords_out, float *feats_out, void *stream)
{
    const EpMap *m = reinterpret_cast<const EpMap *>(handle);
    if (!m || (m->size > 0 && (!coords_out || !feats_out))) return EPRECON_ERR_ARG;
    if (m->size == 0) return EPRECON_OK;
    hipStream_t st = (hipStream_t)stream;
    EP_HIP_CHECK(hipMemcpyAsync(coords_out, m->coords[m->cur], (size_t)m->size * 3 * sizeof(int32_t),
                                hipMemcpyDeviceToDevice, st));
    EP_HIP_CHECK(hipMemcpyAsync(feats_out, m->feats[m->cur], (size_t)m->size * m->channels * sizeof(float),
                                hipMemcpyDeviceToDevice, st));
    return EPRECON_OK;
}

int eprecon_map_import_async(void *handle, const int32_t *coords, const float *feats, int64_t n, void *stream)
{
    EpMap *m = as_map(handle);
    if (!m || n < 0 || (n > 0 && (!coords || !feats))) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    EP_HIP_CHECK(hipStreamSynchronize(st));  // a reallocation below must not free rows still being read
    m->size = 0;
    int rc = ensure_rows(m, n);
    if (rc != EPRECON_OK) return rc;
    if (n > 0) {
        EP_HIP_CHECK(hipMemcpyAsync(m->coords[m->cur], coords, (size_t)n * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        EP_HIP_CHECK(hipMemcpyAsync(m->feats[m->cur], feats, (size_t)n * m->channels * sizeof(float),
                                    hipMemcpyDeviceToDevice, st));
        EP_HIP_CHECK(hipMemsetAsync(m->stamps[m->cur], 0, (size_t)n * sizeof(int32_t), st));  // origin unknown
    }
    m->size = n;
    m->kept = -1;
    return EPRECON_OK;
}

int eprecon_map_set_fragment(void *handle, int fragment_index)
{
    EpMap *m = as_map(handle);
    if (!m || fragment_index < -1 || fragment_index > 0x3ffffffe) return EPRECON_ERR_ARG;
    m->fuse_stamp = fragment_index + 1;  // -1: rows appended by update carry no origin (single-GPU default)
    return EPRECON_OK;
}

int eprecon_map_stamps_async(void *handle, int32_t *export_to, const int32_t *import_from, int fill_all, int32_t fill_value,
                             void *stream)
{
    EpMap *m = as_map(handle);
    if (!m) return EPRECON_ERR_ARG;
    if (m->size == 0) return EPRECON_OK;
    hipStream_t st = (hipStream_t)stream;
    if (import_from)
        EP_HIP_CHECK(hipMemcpyAsync(m->stamps[m->cur], import_from, (size_t)m->size * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (fill_all) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(m->size, 256)), dim3(256), 0, st, m->stamps[m->cur], m->size,
                           fill_value);
        EP_LAUNCH_CHECK();
    }
    if (export_to)
        EP_HIP_CHECK(hipMemcpyAsync(export_to, m->stamps[m->cur], (size_t)m->size * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return EPRECON_OK;
}

int eprecon_map_select_boundary_async(void *handle, const int32_t *boxes_lo, int n_boxes, int own_box, int dim,
                                      int32_t *count_out, void *stream)
{
    EpMap *m = as_map(handle);
    if (!m || n_boxes < 0 || n_boxes > 4096 || dim <= 0 || !count_out || (n_boxes > 0 && !boxes_lo)) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    m->n_selected = -1;
    if (m->size == 0 || n_boxes == 0) {
        EP_HIP_CHECK(hipMemsetAsync(count_out, 0, sizeof(int32_t), st));
        m->n_selected = 0;
        return EPRECON_OK;
    }
    int rc = ensure_sel(m, m->size);
    if (rc != EPRECON_OK) return rc;
    hipLaunchKernelGGL(map_select_kernel, dim3((unsigned)ceil_div(m->size, 256)), dim3(256), 0, st,
                       (const int32_t *)m->coords[m->cur], (const int32_t *)m->stamps[m->cur], (int)m->size, boxes_lo, n_boxes,
                       own_box, dim, m->sel);
    EP_LAUNCH_CHECK();
    return ep::exclusive_scan_i32(m->sel, (int)m->size, m->sel_rank, m->sel_scratch, count_out, st);
}

int eprecon_map_pack_boundary_async(void *handle, float *payload, int64_t n_rows, void *stream)
{
    EpMap *m = as_map(handle);
    if (!m || n_rows < 0 || (n_rows > 0 && !payload)) return EPRECON_ERR_ARG;
    if (n_rows == 0 || m->size == 0) return EPRECON_OK;
    if (m->n_selected == 0) return EPRECON_ERR_ARG;  // nothing was selected, yet rows are asked for
    const int W = m->channels + 4;
    hipLaunchKernelGGL(map_pack_kernel, dim3((unsigned)ceil_div(m->size * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t *)m->sel, (const int32_t *)m->sel_rank, (int)m->size, (const int32_t *)m->coords[m->cur],
                       (const float *)m->feats[m->cur], (const int32_t *)m->stamps[m->cur], m->channels, payload);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_map_merge_boundary(void *handle, const float *payload, int64_t n_rows, const int32_t *box_lo_host, int dim,
                               int64_t *n_added_host, void *stream)
{
    EpMap *m = as_map(handle);
    if (!m || n_rows < 0 || dim <= 0 || dim > 512 || !box_lo_host || (n_rows > 0 && !payload)) return EPRECON_ERR_ARG;
    if (n_added_host) *n_added_host = 0;
    if (n_rows == 0) return EPRECON_OK;
    hipStream_t st = (hipStream_t)stream;
    const int cells = dim * dim * dim;
    if (m->size + n_rows > m->cap) {
        EP_HIP_CHECK(hipStreamSynchronize(st));
        int rc = ensure_rows(m, m->size + n_rows);
        if (rc != EPRECON_OK) return rc;
    }
    int rc = ensure_dense(m, dim);
    if (rc == EPRECON_OK) rc = ensure_sel(m, n_rows);
    if (rc != EPRECON_OK) return rc;
    const size_t seg = align_up((size_t)cells * 4, 256);
    int32_t *idx = reinterpret_cast<int32_t *>(m->dense);          // local row of a cell, -1 = none
    int32_t *best = reinterpret_cast<int32_t *>(m->dense + seg);   // newest received fragment + 1, 0 = none
    EP_HIP_CHECK(hipMemsetAsync(idx, 0xFF, seg, st));
    EP_HIP_CHECK(hipMemsetAsync(best, 0, seg, st));
    const int lx = box_lo_host[0], ly = box_lo_host[1], lz = box_lo_host[2], W = m->channels + 4;
    const dim3 blk(256), grows((unsigned)ceil_div(n_rows, 256));
    if (m->size > 0) {
        hipLaunchKernelGGL(map_index_kernel, dim3((unsigned)ceil_div(m->size, 256)), blk, 0, st, (const int32_t *)m->coords[m->cur],
                           (int)m->size, dim, lx, ly, lz, idx);
        EP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(merge_best_kernel, grows, blk, 0, st, payload, (int)n_rows, W, dim, lx, ly, lz, best);
    EP_LAUNCH_CHECK();
    int32_t *action = m->sel_rank, *flag = m->sel;   // (selection scratch: the send side of this exchange is over)
    hipLaunchKernelGGL(merge_claim_kernel, grows, blk, 0, st, payload, (int)n_rows, W, dim, lx, ly, lz, best, (const int32_t *)idx,
                       (const int32_t *)m->stamps[m->cur], action);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(action_to_flag_kernel, grows, blk, 0, st, (const int32_t *)action, (int)n_rows, flag);
    EP_LAUNCH_CHECK();
    int32_t *add_rank = m->sel_aux;  // ranks of the appended rows, payload order
    rc = ep::exclusive_scan_i32(flag, (int)n_rows, add_rank, m->sel_scratch, m->counts_dev + 2, st);
    if (rc != EPRECON_OK) return rc;
    hipLaunchKernelGGL(merge_apply_kernel, dim3((unsigned)ceil_div(n_rows * W, 256)), blk, 0, st, payload, (int)n_rows, m->channels,
                       (const int32_t *)action, (const int32_t *)add_rank, m->size, m->coords[m->cur], m->feats[m->cur],
                       m->stamps[m->cur]);
    EP_LAUNCH_CHECK();
    EP_HIP_CHECK(hipMemcpyAsync(m->counts_host + 2, m->counts_dev + 2, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    EP_HIP_CHECK(hipStreamSynchronize(st));
    m->size += m->counts_host[2];
    m->kept = -1;
    if (n_added_host) *n_added_host = m->counts_host[2];
    return EPRECON_OK;
}

int eprecon_map_crop_union(void *handle, const int32_t *cur_coords, const float *cur_feat, int64_t n_cur, int ld_cur,
                           int dim, int interval, int activity_mode, const int32_t *relative_origin_host,
                           int32_t *updated, int32_t *src_cur, int32_t *src_glob, int64_t *counts_host, void *stream)
{
    EpMap *m = as_map(handle);
    if (!m || n_cur < 0 || dim <= 0 || dim > 512 || interval <= 0 || !relative_origin_host || !updated || !src_cur ||
        !src_glob || !counts_host || (n_cur > 0 && (!cur_coords || !cur_feat || ld_cur < m->channels)))
        return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int cells = dim * dim * dim;
    int rc = ensure_dense(m, dim);
    if (rc == EPRECON_OK) rc = ensure_flags(m, m->size, cells);
    if (rc != EPRECON_OK) return rc;
    const size_t seg = align_up((size_t)cells * 4, 256);
    CropParams p;
    p.cur_coords = cur_coords; p.cur_feat = cur_feat; p.n_cur = (int)n_cur; p.ld_cur = ld_cur;
    p.glob_coords = m->coords[m->cur]; p.glob_feat = m->feats[m->cur]; p.n_glob = (int)m->size;
    p.C = m->channels; p.D = dim; p.interval = interval; p.mode = activity_mode;
    for (int a = 0; a < 3; ++a) p.rel[a] = m->rel[a] = relative_origin_host[a];
    p.idx_cur = reinterpret_cast<int32_t *>(m->dense);
    p.idx_glob = reinterpret_cast<int32_t *>(m->dense + seg);
    p.flag = reinterpret_cast<int32_t *>(m->dense + 2 * seg);
    int32_t *rank = reinterpret_cast<int32_t *>(m->dense + 3 * seg);
    p.keep = m->keep;
    EP_HIP_CHECK(hipMemsetAsync(p.idx_cur, 0xFF, 2 * seg, st));
    EP_HIP_CHECK(hipMemsetAsync(p.flag, 0, seg, st));
    const int64_t rows = n_cur + m->size;
    if (rows > 0) {
        hipLaunchKernelGGL(map_crop_scatter_kernel, dim3((unsigned)ceil_div(rows, 32)), dim3(256), 0, st, p);
        EP_LAUNCH_CHECK();
    }
    rc = ep::exclusive_scan_i32(p.flag, cells, rank, m->scan_scratch, m->counts_dev, st);
    if (rc != EPRECON_OK) return rc;
    rc = ep::exclusive_scan_i32(m->keep, (int)m->size, m->keep_rank, m->scan_scratch + m->scratch_cap / 2,
                                m->counts_dev + 1, st);
    if (rc != EPRECON_OK) return rc;
    hipLaunchKernelGGL(map_emit_kernel, dim3((unsigned)ceil_div(cells, 256)), dim3(256), 0, st, (const int32_t *)p.flag,
                       (const int32_t *)rank, (const int32_t *)p.idx_cur, (const int32_t *)p.idx_glob, dim, updated,
                       src_cur, src_glob);
    EP_LAUNCH_CHECK();
    EP_HIP_CHECK(hipMemcpyAsync(m->counts_host, m->counts_dev, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    EP_HIP_CHECK(hipStreamSynchronize(st));
    m->kept = m->counts_host[1];
    counts_host[0] = m->counts_host[0];
    counts_host[1] = m->size - m->kept;
    return EPRECON_OK;
}

int eprecon_map_gather_async(const void *handle, const int32_t *src_glob, int64_t n, int col0, int channels, float fill,
                             float *out, int ld_out, void *stream)
{
    const EpMap *m = reinterpret_cast<const EpMap *>(handle);
    if (!m || n < 0 || channels <= 0 || col0 < 0 || col0 + channels > m->channels || ld_out < channels ||
        (n > 0 && (!src_glob || !out)))
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(map_gather_kernel, dim3((unsigned)ceil_div(n * channels, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)m->feats[m->cur], m->channels, col0, src_glob, (int)n, channels, fill, out, ld_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_map_update_async(void *handle, const int32_t *updated, int64_t n, const float *values, int ld_values,
                             void *stream)
{
    EpMap *m = as_map(handle);
    if (!m || n < 0 || (n > 0 && (!updated || !values || ld_values < m->channels))) return EPRECON_ERR_ARG;
    if (m->kept < 0) return EPRECON_ERR_ARG;  // no crop since the last update / import, or its count was not committed yet
    hipStream_t st = (hipStream_t)stream;
    const int64_t new_size = m->kept + n;
    if (new_size > m->cap) {
        EP_HIP_CHECK(hipStreamSynchronize(st));
        int rc = ensure_rows(m, new_size);
        if (rc != EPRECON_OK) return rc;
    }
    const int src = m->cur, dst = 1 - m->cur;
    const int W = m->channels + 3;
    if (m->size > 0) {
        hipLaunchKernelGGL(map_compact_kernel, dim3((unsigned)ceil_div(m->size * W, 256)), dim3(256), 0, st,
                           (const int32_t *)m->keep, (const int32_t *)m->keep_rank, (int)m->size,
                           (const int32_t *)m->coords[src], (const float *)m->feats[src], (const int32_t *)m->stamps[src],
                           m->channels, m->coords[dst], m->feats[dst], m->stamps[dst]);
        EP_LAUNCH_CHECK();
    }
    if (n > 0) {
        hipLaunchKernelGGL(map_append_kernel, dim3((unsigned)ceil_div(n * W, 256)), dim3(256), 0, st, updated, values,
                           ld_values, (int)n, m->channels, m->rel[0], m->rel[1], m->rel[2], m->kept, m->fuse_stamp,
                           m->coords[dst], m->feats[dst], m->stamps[dst]);
        EP_LAUNCH_CHECK();
    }
    m->cur = dst;
    m->size = new_size;
    m->kept = -1;
    return EPRECON_OK;
}

int eprecon_map_target_fuse(void *handle, const float *tsdf_gt, const uint8_t *occ_gt, int dim,
                            const int32_t *relative_origin_host, const int32_t *updated, int64_t n, float *tsdf_target_out,
                            void *stream)
{
    EpMap *m = as_map(handle);
    if (!m || m->channels != 1 || !tsdf_gt || !occ_gt || dim <= 0 || dim > 512 || !relative_origin_host || n < 0 ||
        (n > 0 && (!updated || !tsdf_target_out)))
        return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int cells = dim * dim * dim;
    int rc = ensure_dense(m, dim);
    if (rc == EPRECON_OK) rc = ensure_flags(m, m->size, cells);
    if (rc != EPRECON_OK) return rc;
    const size_t seg = align_up((size_t)cells * 4, 256);
    int32_t *flag = reinterpret_cast<int32_t *>(m->dense + 2 * seg);
    int32_t *rank = reinterpret_cast<int32_t *>(m->dense + 3 * seg);
    float *vol = reinterpret_cast<float *>(m->dense + 4 * seg);
    const int rx = relative_origin_host[0], ry = relative_origin_host[1], rz = relative_origin_host[2];
    const dim3 blk(256), gcells((unsigned)ceil_div(cells, 256));
    hipLaunchKernelGGL(fill_f32_kernel, gcells, blk, 0, st, vol, cells, 1.0f);
    EP_LAUNCH_CHECK();
    if (m->size > 0) {
        hipLaunchKernelGGL(target_scatter_kernel, dim3((unsigned)ceil_div(m->size, 256)), blk, 0, st,
                           (const int32_t *)m->coords[m->cur], (const float *)m->feats[m->cur], (int)m->size, dim, rx, ry,
                           rz, vol, m->keep);
        EP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(target_merge_kernel, gcells, blk, 0, st, tsdf_gt, occ_gt, cells, vol, flag);
    EP_LAUNCH_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(target_lookup_kernel, dim3((unsigned)ceil_div(n, 256)), blk, 0, st, (const float *)vol, updated,
                           (int)n, dim, tsdf_target_out);
        EP_LAUNCH_CHECK();
    }
    rc = ep::exclusive_scan_i32(flag, cells, rank, m->scan_scratch, m->counts_dev, st);
    if (rc != EPRECON_OK) return rc;
    rc = ep::exclusive_scan_i32(m->keep, (int)m->size, m->keep_rank, m->scan_scratch + m->scratch_cap / 2,
                                m->counts_dev + 1, st);
    if (rc != EPRECON_OK) return rc;
    EP_HIP_CHECK(hipMemcpyAsync(m->counts_host, m->counts_dev, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    EP_HIP_CHECK(hipStreamSynchronize(st));
    const int64_t n_new = m->counts_host[0], kept = m->size > 0 ? m->counts_host[1] : 0;
    rc = ensure_rows(m, kept + n_new);
    if (rc != EPRECON_OK) return rc;
    const int src = m->cur, dst = 1 - m->cur;
    if (m->size > 0) {
        hipLaunchKernelGGL(map_compact_kernel, dim3((unsigned)ceil_div(m->size * 4, 256)), blk, 0, st,
                           (const int32_t *)m->keep, (const int32_t *)m->keep_rank, (int)m->size,
                           (const int32_t *)m->coords[src], (const float *)m->feats[src], (const int32_t *)m->stamps[src], 1,
                           m->coords[dst], m->feats[dst], m->stamps[dst]);
        EP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(target_append_kernel, gcells, blk, 0, st, (const float *)vol, (const int32_t *)flag,
                       (const int32_t *)rank, dim, rx, ry, rz, kept, m->coords[dst], m->feats[dst], m->stamps[dst]);
    EP_LAUNCH_CHECK();
    m->cur = dst;
    m->size = kept + n_new;
    m->kept = -1;
    return EPRECON_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * One GRU-fusion level with device-side counts (include/eprecon_hip.h: eprecon_gru_stage_desc)
 * ------------------------------------------------------------------------------------------------------------------ */
int64_t eprecon_gru_stage_capacity(const void *map, int64_t n_cur, int dim)
{
    const EpMap *m = reinterpret_cast<const EpMap *>(map);
    if (!m || n_cur < 0 || dim <= 0) return -1;
    const int64_t cells = (int64_t)dim * dim * dim;
    const int64_t cap = n_cur + m->size < cells ? n_cur + m->size : cells;
    return cap > 0 ? cap : 1;
}

size_t eprecon_gru_stage_workspace_bytes(int64_t capacity)
{
    const int64_t cap = capacity > 0 ? capacity : 1;
    return 2 * align_up((size_t)cap * sizeof(int32_t), 256) + eprecon_unique_workspace_bytes(cap);
}

int eprecon_gru_stage_begin_async(const eprecon_gru_stage_desc *d, void *stream)
{
    if (!d || !d->map || d->n_cur < 0 || d->dim <= 0 || d->dim > 512 || d->interval <= 0 || d->capacity <= 0 || !d->updated ||
        !d->out_coords || !d->r_coords || !d->hx_voxel || !d->hx_image || !d->counts || !d->origin || !d->w2ac || !d->workspace ||
        !(d->resolution > 0.0f) || !d->scaled1 || !d->vox1 || !d->inverse1 || !d->uniq1 || !d->table1 || !d->scaled2 || !d->vox2 ||
        !d->inverse2 || !d->uniq2 || !d->table2)
        return EPRECON_ERR_ARG;
    EpMap *m = as_map(d->map);
    EpMap *tm = as_map(d->target_map);
    const int C = m->channels;
    if (d->ch_voxel <= 0 || d->ch_voxel >= C || (d->n_cur > 0 && (!d->cur_coords || !d->cur_feat || d->ld_cur < C))) return EPRECON_ERR_ARG;
    if (d->capacity < eprecon_gru_stage_capacity(d->map, d->n_cur, d->dim)) return EPRECON_ERR_ARG;
    if (d->workspace_bytes < eprecon_gru_stage_workspace_bytes(d->capacity)) return EPRECON_ERR_WORKSPACE;
    if (tm && (tm->channels != 1 || !d->tsdf_gt || !d->occ_gt || !d->tsdf_target)) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int dim = d->dim, cells = dim * dim * dim;
    const int cap = (int)d->capacity;
    int rc = ensure_dense(m, dim);
    if (rc == EPRECON_OK) rc = ensure_flags(m, m->size, cells);
    if (rc != EPRECON_OK) return rc;
    char *ws = reinterpret_cast<char *>(d->workspace);
    const size_t iseg = align_up((size_t)cap * sizeof(int32_t), 256);
    int32_t *src_cur = reinterpret_cast<int32_t *>(ws);
    int32_t *src_glob = reinterpret_cast<int32_t *>(ws + iseg);
    void *uws = ws + 2 * iseg;
    const size_t uws_bytes = d->workspace_bytes - 2 * iseg;
    int32_t *cnt = d->counts;
    const size_t seg = align_up((size_t)cells * 4, 256);
    if ((reinterpret_cast<uintptr_t>(cnt) & 15) != 0) return EPRECON_ERR_ARG;

    // --- everything the call has to reset, in ONE launch (was: three memsets, a fill kernel and two table clears): the
    //     counters, the two index volumes (-1) and the flag volume (0) of the crop, the ground-truth twin's dense volume (1.0)
    //     and the two hash tables of the shared voxelisations ---
    {
        ep::FillRegion reg[ep::kMaxFillRegions];
        int nr = 0;
        reg[nr++] = ep::FillRegion{cnt, 8 * sizeof(int32_t), 0u};
        reg[nr++] = ep::FillRegion{m->dense, 2 * seg, 0xFFFFFFFFu};
        reg[nr++] = ep::FillRegion{m->dense + 2 * seg, seg, 0u};
        if (tm) {
            rc = ensure_dense(tm, dim);
            if (rc == EPRECON_OK) rc = ensure_flags(tm, tm->size, cells);
            if (rc != EPRECON_OK) return rc;
            reg[nr++] = ep::FillRegion{tm->dense + 4 * seg, seg, 0x3f800000u};   // 1.0f (the whole 256-byte-aligned segment)
        }
        rc = ep::table_clear_regions(d->table1, d->table_capacity, reg + nr);
        if (rc != EPRECON_OK) return rc;
        nr += 3;
        rc = ep::table_clear_regions(d->table2, d->table_capacity, reg + nr);
        if (rc != EPRECON_OK) return rc;
        nr += 3;
        rc = ep::multi_fill(reg, nr, st);
        if (rc != EPRECON_OK) return rc;
    }

    // --- crop + union (the kernels of eprecon_map_crop_union, without its host read) ---
    CropParams p;
    p.cur_coords = d->cur_coords; p.cur_feat = d->cur_feat; p.n_cur = (int)d->n_cur; p.ld_cur = d->ld_cur;
    p.glob_coords = m->coords[m->cur]; p.glob_feat = m->feats[m->cur]; p.n_glob = (int)m->size;
    p.C = C; p.D = dim; p.interval = d->interval; p.mode = d->activity_mode;
    for (int a = 0; a < 3; ++a) p.rel[a] = m->rel[a] = d->rel[a];
    p.idx_cur = reinterpret_cast<int32_t *>(m->dense);
    p.idx_glob = reinterpret_cast<int32_t *>(m->dense + seg);
    p.flag = reinterpret_cast<int32_t *>(m->dense + 2 * seg);
    int32_t *rank = reinterpret_cast<int32_t *>(m->dense + 3 * seg);
    p.keep = m->keep;
    const int64_t rows = d->n_cur + m->size;
    if (rows > 0) {
        hipLaunchKernelGGL(map_crop_scatter_kernel, dim3((unsigned)ceil_div(rows, 32)), dim3(256), 0, st, p);
        EP_LAUNCH_CHECK();
    }
    rc = ep::exclusive_scan_i32(p.flag, cells, rank, m->scan_scratch, cnt + 0, st);
    if (rc != EPRECON_OK) return rc;
    rc = ep::exclusive_scan_i32(m->keep, (int)m->size, m->keep_rank, m->scan_scratch + m->scratch_cap / 2, cnt + 1, st);
    if (rc != EPRECON_OK) return rc;
    hipLaunchKernelGGL(map_emit_kernel, dim3((unsigned)ceil_div(cells, 256)), dim3(256), 0, st, (const int32_t *)p.flag,
                       (const int32_t *)rank, (const int32_t *)p.idx_cur, (const int32_t *)p.idx_glob, dim, d->updated, src_cur, src_glob);
    EP_LAUNCH_CHECK();
    m->kept = -2;   // a crop is pending; eprecon_gru_stage_commit supplies the count the host read
    const int32_t *n_u = cnt + 0;

    // --- [h | x] rows of the two cells ---
    hipLaunchKernelGGL(stage_gather_kernel, dim3((unsigned)ceil_div((int64_t)cap * C, 256)), dim3(256), 0, st,
                       (const float *)m->feats[m->cur], d->cur_feat, d->ld_cur, (const int32_t *)src_glob, (const int32_t *)src_cur, cap,
                       n_u, C, d->ch_voxel, d->hx_voxel, d->hx_image);
    EP_LAUNCH_CHECK();

    // --- ground-truth twin: dense volume <- map rows inside the FBV <- the fragment's ground truth; targets at the union ---
    if (tm) {
        int32_t *tflag = reinterpret_cast<int32_t *>(tm->dense + 2 * seg);
        int32_t *trank = reinterpret_cast<int32_t *>(tm->dense + 3 * seg);
        float *vol = reinterpret_cast<float *>(tm->dense + 4 * seg);     // (filled with 1.0 by the call's first launch)
        for (int a = 0; a < 3; ++a) tm->rel[a] = d->rel[a];
        const dim3 blk(256), gcells((unsigned)ceil_div(cells, 256));
        if (tm->size > 0) {
            hipLaunchKernelGGL(target_scatter_kernel, dim3((unsigned)ceil_div(tm->size, 256)), blk, 0, st,
                               (const int32_t *)tm->coords[tm->cur], (const float *)tm->feats[tm->cur], (int)tm->size, dim, d->rel[0],
                               d->rel[1], d->rel[2], vol, tm->keep);
            EP_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(target_merge_kernel, gcells, blk, 0, st, d->tsdf_gt, d->occ_gt, cells, vol, tflag);
        EP_LAUNCH_CHECK();
        hipLaunchKernelGGL(target_lookup_dn_kernel, dim3((unsigned)ceil_div(cap, 256)), blk, 0, st, (const float *)vol,
                           (const int32_t *)d->updated, cap, n_u, dim, d->tsdf_target);
        EP_LAUNCH_CHECK();
        rc = ep::exclusive_scan_i32(tflag, cells, trank, tm->scan_scratch, cnt + 4, st);
        if (rc != EPRECON_OK) return rc;
        rc = ep::exclusive_scan_i32(tm->keep, (int)tm->size, tm->keep_rank, tm->scan_scratch + tm->scratch_cap / 2, cnt + 5, st);
        if (rc != EPRECON_OK) return rc;
        tm->kept = -2;
        tm->pending_dim = dim;
    }

    // --- the fragment's points and the two voxelisations the six SConv3d of the scale share ---
    // (one launch for the points and both quantisations; the tables were reset by the call's first launch; each numbering's
    // last launch leaves its table's status word next to the counts: one host read for everything)
    hipLaunchKernelGGL(stage_points_quantize_kernel, dim3((unsigned)ceil_div(cap, 256)), dim3(256), 0, st, (const int32_t *)d->updated,
                       cap, n_u, d->interval, d->batch_index, d->origin, d->voxel_size, d->w2ac, d->resolution,
                       reinterpret_cast<int4 *>(d->out_coords), reinterpret_cast<float4 *>(d->r_coords),
                       reinterpret_cast<float4 *>(d->scaled1), reinterpret_cast<int4 *>(d->vox1),
                       reinterpret_cast<float4 *>(d->scaled2), reinterpret_cast<int4 *>(d->vox2));
    EP_LAUNCH_CHECK();
    rc = ep::unique_coords_dn(d->vox1, cap, n_u, 1, d->table1, d->table_capacity, d->inverse1, d->uniq1, cnt + 2, uws, uws_bytes,
                              true, cnt + 6, stream);
    if (rc != EPRECON_OK) return rc;
    rc = ep::unique_coords_dn(d->vox2, cap, n_u, 1, d->table2, d->table_capacity, d->inverse2, d->uniq2, cnt + 3, uws, uws_bytes,
                              true, cnt + 7, stream);
    if (rc != EPRECON_OK) return rc;
    return EPRECON_OK;
}

int eprecon_gru_stage_commit_async(void *map, void *target_map, const int32_t *counts_host, void *stream)
{
    EpMap *m = as_map(map);
    EpMap *tm = as_map(target_map);
    if (!m || !counts_host || m->kept != -2) return EPRECON_ERR_ARG;
    if (counts_host[1] < 0 || counts_host[1] > m->size) return EPRECON_ERR_ARG;
    m->kept = counts_host[1];
    if (!tm) return EPRECON_OK;
    if (tm->kept != -2) return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_new = counts_host[4], kept = tm->size > 0 ? counts_host[5] : 0;
    // (the dims of the pending crop are those of the dense workspace the begin call filled)
    if (n_new < 0 || kept < 0 || kept > tm->size) return EPRECON_ERR_ARG;
    if (kept + n_new > tm->cap) {
        EP_HIP_CHECK(hipStreamSynchronize(st));
        int rc = ensure_rows(tm, kept + n_new);
        if (rc != EPRECON_OK) return rc;
    }
    const int dim = tm->pending_dim, cells = dim * dim * dim;
    const size_t seg = align_up((size_t)cells * 4, 256);
    int32_t *tflag = reinterpret_cast<int32_t *>(tm->dense + 2 * seg);
    int32_t *trank = reinterpret_cast<int32_t *>(tm->dense + 3 * seg);
    float *vol = reinterpret_cast<float *>(tm->dense + 4 * seg);
    const dim3 blk(256), gcells((unsigned)ceil_div(cells, 256));
    const int src = tm->cur, dst = 1 - tm->cur;
    if (tm->size > 0) {
        hipLaunchKernelGGL(map_compact_kernel, dim3((unsigned)ceil_div(tm->size * 4, 256)), blk, 0, st,
                           (const int32_t *)tm->keep, (const int32_t *)tm->keep_rank, (int)tm->size,
                           (const int32_t *)tm->coords[src], (const float *)tm->feats[src], (const int32_t *)tm->stamps[src], 1,
                           tm->coords[dst], tm->feats[dst], tm->stamps[dst]);
        EP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(target_append_kernel, gcells, blk, 0, st, (const float *)vol, (const int32_t *)tflag,
                       (const int32_t *)trank, dim, tm->rel[0], tm->rel[1], tm->rel[2], kept, tm->coords[dst], tm->feats[dst],
                       tm->stamps[dst]);
    EP_LAUNCH_CHECK();
    tm->cur = dst;
    tm->size = kept + n_new;
    tm->kept = -1;
    return EPRECON_OK;
}

}  // extern "C"
