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
    int64_t kept = -1;  // rows outside the FBV of the last crop (-1: no crop pending)
    int rel[3] = {0, 0, 0};
};

int ensure_rows(EpMap *m, int64_t rows)
{
    if (rows <= m->cap) return EPRECON_OK;
    int64_t cap = m->cap > 0 ? m->cap : 4096;
    while (cap < rows) cap *= 2;
    for (int b = 0; b < 2; ++b) {
        int32_t *c = nullptr;
        float *f = nullptr;
        EP_HIP_CHECK(hipMalloc(&c, (size_t)cap * 3 * sizeof(int32_t)));
        EP_HIP_CHECK(hipMalloc(&f, (size_t)cap * m->channels * sizeof(float)));
        if (b == m->cur && m->size > 0) {  // only the live buffer carries data
            EP_HIP_CHECK(hipMemcpy(c, m->coords[b], (size_t)m->size * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice));
            EP_HIP_CHECK(hipMemcpy(f, m->feats[b], (size_t)m->size * m->channels * sizeof(float),
                                   hipMemcpyDeviceToDevice));
        }
        if (m->coords[b]) EP_HIP_CHECK(hipFree(m->coords[b]));
        if (m->feats[b]) EP_HIP_CHECK(hipFree(m->feats[b]));
        m->coords[b] = c;
        m->feats[b] = f;
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
                                                          const int32_t *c_in, const float *f_in, int C,
                                                          int32_t *c_out, float *f_out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int W = C + 3;
    if (e >= (int64_t)n * W) return;
    const int j = (int)(e / W), c = (int)(e - (int64_t)j * W);
    if (!keep[j]) return;
    const int o = keep_rank[j];
    if (c < 3)
        c_out[3 * (size_t)o + c] = c_in[3 * (size_t)j + c];
    else
        f_out[(size_t)o * C + (c - 3)] = f_in[(size_t)j * C + (c - 3)];
}

__global__ __launch_bounds__(256) void map_append_kernel(const int32_t *updated, const float *values, int ld_v, int n, int C,
                                                         int rx, int ry, int rz, int64_t base, int32_t *c_out, float *f_out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int W = C + 3;
    if (e >= (int64_t)n * W) return;
    const int i = (int)(e / W), c = (int)(e - (int64_t)i * W);
    const size_t o = (size_t)(base + i);
    if (c < 3)
        c_out[3 * o + c] = updated[3 * (size_t)i + c] + (c == 0 ? rx : (c == 1 ? ry : rz));
    else
        f_out[o * C + (c - 3)] = values[(size_t)i * ld_v + (c - 3)];
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
                                     int rz, int64_t base, int32_t *c_out, float *f_out)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= D * D * D || !flag[cell]) return;
    const size_t o = (size_t)(base + rank[cell]);
    c_out[3 * o + 0] = cell / (D * D) + rx;
    c_out[3 * o + 1] = (cell / D) % D + ry;
    c_out[3 * o + 2] = cell % D + rz;
    f_out[o] = vol[cell];
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
    }
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
    }
    m->size = n;
    m->kept = -1;
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
    if (m->kept < 0) return EPRECON_ERR_ARG;  // no crop_union since the last update / import
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
                           (const int32_t *)m->coords[src], (const float *)m->feats[src], m->channels, m->coords[dst],
                           m->feats[dst]);
        EP_LAUNCH_CHECK();
    }
    if (n > 0) {
        hipLaunchKernelGGL(map_append_kernel, dim3((unsigned)ceil_div(n * W, 256)), dim3(256), 0, st, updated, values,
                           ld_values, (int)n, m->channels, m->rel[0], m->rel[1], m->rel[2], m->kept, m->coords[dst],
                           m->feats[dst]);
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
                           (const int32_t *)m->coords[src], (const float *)m->feats[src], 1, m->coords[dst], m->feats[dst]);
        EP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(target_append_kernel, gcells, blk, 0, st, (const float *)vol, (const int32_t *)flag,
                       (const int32_t *)rank, dim, rx, ry, rz, kept, m->coords[dst], m->feats[dst]);
    EP_LAUNCH_CHECK();
    m->cur = dst;
    m->size = kept + n_new;
    m->kept = -1;
    return EPRECON_OK;
}

}  // extern "C"
