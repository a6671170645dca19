// GRU-fusion bookkeeping on gfx950: union of the current fragment's voxels with the part of the
// persistent global map that falls inside the fragment bounding volume (FBV), in raster order.
//
// Replaces GRUFusion.convert2dense + the dense gathers of GRUFusion.forward
// (models/gru_fusion.py:67-114,321-326 of the reference; utils.py:169-180 sparse_to_dense_channel).
// The reference materialises two dense [D,D,D,C] feature volumes (170 MB each at the finest
// scale), takes torch.nonzero of "any channel != 0" and gathers both volumes at the result.  Here
// only two int32 INDEX volumes [D^3] are scattered; activity flags are scanned into raster-order
// ranks and the feature rows are gathered straight from the sparse inputs — same output, same
// order, including the reference's quirk that a row whose every channel is exactly 0.0 does not
// activate its voxel (models/gru_fusion.py:96).
#include "common.hpp"

namespace {
using namespace ep;

struct UnionParams {
    const int32_t *cur_coords;  // [n_cur,4] (b,x,y,z) finest units, one batch element
    const float *cur_feat;      // [n_cur, ld_cur]
    int n_cur, ld_cur;
    const int32_t *glob_coords;  // [n_glob,3] scene-grid units of this scale
    const float *glob_feat;      // [n_glob, ld_glob]
    int n_glob, ld_glob;
    int C, D, interval, mode;
    int rel[3];
    int32_t *idx_cur, *idx_glob, *flag;  // [D^3]
    uint8_t *glob_valid;                 // [n_glob] inside the FBV
};

// 8 lanes per row: does the row activate its voxel?  mode 0 (features): any channel != 0;
// mode 1 (TSDF, direct substitution): any |channel| < 1   (models/gru_fusion.py:94-96)
__device__ __forceinline__ bool row_nonzero(const float *row, int C, int g, int mode)
{
    bool nz = false;
    for (int c = g; c < C; c += 8) nz |= mode ? (fabsf(row[c]) < 1.0f) : (row[c] != 0.0f);
    unsigned long long m = __ballot(nz);
    const int lane = threadIdx.x & 63;
    return ((m >> (lane & ~7)) & 0xFFull) != 0ull;
}

__global__ __launch_bounds__(256) void union_scatter_kernel(UnionParams p)
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
            row = p.glob_feat + (size_t)j * p.ld_glob;
        }
    }
    const bool inside = live && x >= 0 && x < p.D && y >= 0 && y < p.D && z >= 0 && z < p.D;
    const bool nz = row_nonzero(row, (live ? p.C : 0), g, p.mode);
    if (live && g == 0) {
        if (!is_cur) p.glob_valid[j] = inside ? 1 : 0;
        if (inside) {
            const int cell = (x * p.D + y) * p.D + z;
            (is_cur ? p.idx_cur : p.idx_glob)[cell] = j;
            if (nz) p.flag[cell] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void union_emit_kernel(const int32_t *flag, const int32_t *rank,
                                                         const int32_t *idx_cur, const int32_t *idx_glob,
                                                         int D, int32_t *updated, int32_t *src_cur,
                                                         int32_t *src_glob)
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

// out[i, :] = src[i] >= 0 ? feat[src[i], :] : fill
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *feat, int ld_f, const int32_t *src,
                                                          int n, int C, float fill, float *out, int ld_o)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * C) return;
    const int i = (int)(e / C), c = (int)(e - (int64_t)i * C);
    const int j = src[i];
    out[(size_t)i * ld_o + c] = j >= 0 ? feat[(size_t)j * ld_f + c] : fill;
}

}  // namespace

extern "C" {

size_t eprecon_fbv_union_workspace_bytes(int dim)
{
    const size_t cells = (size_t)dim * dim * dim;
    return 4 * align_up(cells * 4, 256) + align_up((size_t)ceil_div((int64_t)cells, 2048) * 4, 256) + 256;
}

/*
 * Outputs (caller-allocated, capacity dim^3 rows each): updated int32[n_out,3] local grid coords in
 * raster order; src_cur / src_glob int32[n_out] = row of the current / global voxel at that cell
 * or -1; glob_valid u8[n_glob] = global voxel lies inside the FBV; n_out_dev int32[1].
 */
int eprecon_fbv_union_async(const int32_t *cur_coords, const float *cur_feat, int64_t n_cur, int ld_cur,
                            const int32_t *glob_coords, const float *glob_feat, int64_t n_glob, int ld_glob,
                            int channels, int dim, int interval, int activity_mode,
                            const int32_t *relative_origin_host, int32_t *updated, int32_t *src_cur, int32_t *src_glob, uint8_t *glob_valid,
                            int32_t *n_out_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n_cur < 0 || n_glob < 0 || channels <= 0 || dim <= 0 || dim > 512 || interval <= 0 ||
        !relative_origin_host || !updated || !src_cur || !src_glob || !n_out_dev || !workspace ||
        (n_cur > 0 && (!cur_coords || !cur_feat)) || (n_glob > 0 && (!glob_coords || !glob_feat || !glob_valid)))
        return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_fbv_union_workspace_bytes(dim)) return EPRECON_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int cells = dim * dim * dim;
    const size_t seg = align_up((size_t)cells * 4, 256);
    char *ws = reinterpret_cast<char *>(workspace);
    UnionParams p;
    p.cur_coords = cur_coords; p.cur_feat = cur_feat; p.n_cur = (int)n_cur; p.ld_cur = ld_cur;
    p.glob_coords = glob_coords; p.glob_feat = glob_feat; p.n_glob = (int)n_glob; p.ld_glob = ld_glob;
    p.C = channels; p.D = dim; p.interval = interval; p.mode = activity_mode;
    p.rel[0] = relative_origin_host[0]; p.rel[1] = relative_origin_host[1]; p.rel[2] = relative_origin_host[2];
    p.idx_cur = reinterpret_cast<int32_t *>(ws);
    p.idx_glob = reinterpret_cast<int32_t *>(ws + seg);
    p.flag = reinterpret_cast<int32_t *>(ws + 2 * seg);
    int32_t *rank = reinterpret_cast<int32_t *>(ws + 3 * seg);
    int32_t *scratch = reinterpret_cast<int32_t *>(ws + 4 * seg);
    p.glob_valid = glob_valid;
    EP_HIP_CHECK(hipMemsetAsync(p.idx_cur, 0xFF, 2 * seg, st));
    EP_HIP_CHECK(hipMemsetAsync(p.flag, 0, seg, st));
    const int64_t rows = n_cur + n_glob;
    if (rows > 0) {
        hipLaunchKernelGGL(union_scatter_kernel, dim3((unsigned)ceil_div(rows, 32)), dim3(256), 0, st, p);
        EP_LAUNCH_CHECK();
    }
    int rc = ep::exclusive_scan_i32(p.flag, cells, rank, scratch, n_out_dev, st);
    if (rc != EPRECON_OK) return rc;
    hipLaunchKernelGGL(union_emit_kernel, dim3((unsigned)ceil_div(cells, 256)), dim3(256), 0, st,
                       (const int32_t *)p.flag, (const int32_t *)rank, (const int32_t *)p.idx_cur,
                       (const int32_t *)p.idx_glob, dim, updated, src_cur, src_glob);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_gather_rows_async(const float *feat, int ld_feat, const int32_t *src, int64_t n, int channels,
                              float fill, float *out, int ld_out, void *stream)
{
    if (n < 0 || channels <= 0 || (n > 0 && (!src || !out)) || ld_out < channels) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(n * channels, 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, ld_feat, src, (int)n, channels, fill, out, ld_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"
