// The geometry of one SPVCNN pass (models/modules.py:75-175 with ops/torchsparse_utils.py:15-105) in ONE call, after the pass's
// host read has given the sizes of the three strided voxel sets: the point -> voxel CSR lists of strides 1 and 4, the two
// k2s2 maps and their transposes, the three 3x3x3 kernel maps, the trilinear corner tables of strides 1 and 4.  Nothing new is
// computed here: the call issues what the Python modules used to issue one by one (14 calls per pass).
#include "common.hpp"

extern "C" {

static size_t geometry_half_bytes(int64_t n, int64_t n1, int64_t n4)
{
    const size_t a = eprecon_segment_workspace_bytes(n, n1), b = eprecon_segment_workspace_bytes(n, n4);
    return ep::align_up(a > b ? a : b, 256);
}

size_t eprecon_spvcnn_geometry_workspace_bytes(int64_t n, int64_t n1, int64_t n4) { return 2 * geometry_half_bytes(n, n1, n4); }

int eprecon_spvcnn_geometry_async(const eprecon_spvcnn_geometry_desc *d, void *stream)
{
    if (!d || d->n < 0 || d->n1 < 0 || d->n2 < 0 || d->n4 < 0) return EPRECON_ERR_ARG;
    if (d->n == 0 || d->n1 == 0) return EPRECON_OK;
    if (d->workspace_bytes < eprecon_spvcnn_geometry_workspace_bytes(d->n, d->n1, d->n4)) return EPRECON_ERR_WORKSPACE;
    const size_t half = geometry_half_bytes(d->n, d->n1, d->n4);
    void *ws_a = d->workspace, *ws_b = (char *)d->workspace + half;
    hipStream_t main = (hipStream_t)stream;
    ep::Fork f;
    int rc = ep::fork_for(main, f);
    if (rc != EPRECON_OK) return rc;
    void *side = (void *)f.side;
#define EP_STEP(call)                      \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != EPRECON_OK) return rc_; \
    } while (0)
    // two independent chains of small launches side by side (see ep::Fork); each is a lambda so that a failing step cannot
    // skip the join, and the first error is what the call returns
    auto strided = [&]() -> int {  // the strided sets: the library's side stream
        if (d->n2 > 0) {
            EP_STEP(eprecon_kernel_map_async(d->table1, d->capacity1, d->coords2, d->n2, 2, 1, d->down12, side));
            EP_STEP(eprecon_transpose_map_async(d->coords1, d->n1, d->parent2, 1, d->up21, side));
            EP_STEP(ep::kernel_map_self_prefilled(d->table2, d->capacity2, d->coords2, d->n2, 2, d->k2, side));
        }
        if (d->n4 > 0) {
            EP_STEP(eprecon_kernel_map_async(d->table2, d->capacity2, d->coords4, d->n4, 2, 2, d->down24, side));
            EP_STEP(eprecon_transpose_map_async(d->coords2, d->n2, d->parent4, 2, d->up42, side));
            EP_STEP(ep::kernel_map_self_prefilled(d->table4, d->capacity4, d->coords4, d->n4, 4, d->k4, side));
            EP_STEP(eprecon_hash_query_async(d->table4, d->capacity4, d->vox, d->n, 4, d->idx4, side));
            EP_STEP(eprecon_segment_lists_async(d->idx4, d->n, d->n4, d->offsets4, d->order4, ws_b, half, side));
            // (corners from the set's kernel map and the points' stride-4 rows: no hash probes, ep::trilinear_from_map)
            EP_STEP(ep::trilinear_from_map(d->scaled, d->n, d->idx4, d->k4, d->n4, 4, d->idx8_4, d->weight8_4, side));
        }
        return EPRECON_OK;
    };
    auto unit = [&]() -> int {  // the stride-1 set: the caller's stream
        EP_STEP(eprecon_segment_lists_async(d->inverse1, d->n, d->n1, d->offsets1, d->order1, ws_a, half, stream));
        EP_STEP(ep::kernel_map_self_prefilled(d->table1, d->capacity1, d->coords1, d->n1, 1, d->k1, stream));
        EP_STEP(ep::trilinear_from_map(d->scaled, d->n, d->inverse1, d->k1, d->n1, 1, d->idx8_1, d->weight8_1, stream));
        return EPRECON_OK;
    };
    {   // the upper halves of the three self maps (mirrored entries are scattered into them: -1 first), ONE launch, before the fork
        ep::FillRegion reg[3] = {ep::kernel_map_self_fill_region(d->k1, d->n1), ep::kernel_map_self_fill_region(d->k2, d->n2),
                                 ep::kernel_map_self_fill_region(d->k4, d->n4)};
        rc = ep::multi_fill(reg, 3, main);
        if (rc != EPRECON_OK) return rc;
    }
    EP_HIP_CHECK(hipEventRecord(f.ev_fork, main));
    EP_HIP_CHECK(hipStreamWaitEvent(f.side, f.ev_fork, 0));
    const int rc_strided = strided();
    EP_HIP_CHECK(hipEventRecord(f.ev_join, f.side));
    const int rc_unit = unit();
    EP_HIP_CHECK(hipStreamWaitEvent(main, f.ev_join, 0));
    if (rc_strided != EPRECON_OK) return rc_strided;
    if (rc_unit != EPRECON_OK) return rc_unit;
#undef EP_STEP
    return EPRECON_OK;
}

}  // extern "C"
