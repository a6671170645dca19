// The rest of a GRU-fusion level's geometry in ONE call, after the level's host read has given the sizes
// (models/gru_fusion.py:332-347 -> models/modules.py:178-222 -> ops/torchsparse_utils.py:15-105): for the two voxelisations the
// six SConv3d of the level's two ConvGRUs share — point -> voxel CSR lists, 3x3x3 kernel maps, trilinear corner tables of the
// first, and for the reference-literal convr the hash order of both sets + the stale-index composition (csrc/hash_order.hip,
// voxelize.hip).  Nothing new is computed here: the call issues what the Python modules used to issue one by one (12 calls
// per level in a stretch of the fragment that is bound by the host's launch rate).
#include "common.hpp"

extern "C" {

size_t eprecon_gru_stage_finish_workspace_bytes(int64_t n, int64_t m1, int64_t m2)
{
    size_t w = eprecon_segment_workspace_bytes(n, m1);
    const size_t cands[] = {eprecon_segment_workspace_bytes(n, m2), eprecon_sphash_order_workspace_bytes(m1),
                            eprecon_sphash_order_workspace_bytes(m2)};
    for (size_t c : cands) w = c > w ? c : w;
    return w;
}

int eprecon_gru_stage_finish_async(const eprecon_gru_finish_desc *d, void *stream)
{
    if (!d || d->n < 0 || d->m1 < 0 || d->m2 < 0) return EPRECON_ERR_ARG;
    if (d->n == 0) return EPRECON_OK;
    if (d->workspace_bytes < eprecon_gru_stage_finish_workspace_bytes(d->n, d->m1, d->m2)) return EPRECON_ERR_WORKSPACE;
    int rc = eprecon_segment_lists_async(d->inverse1, d->n, d->m1, d->offsets1, d->order1, d->workspace, d->workspace_bytes, stream);
    if (rc != EPRECON_OK) return rc;
    rc = eprecon_segment_lists_async(d->inverse2, d->n, d->m2, d->offsets2, d->order2, d->workspace, d->workspace_bytes, stream);
    if (rc != EPRECON_OK) return rc;
    if (d->m1 > 0) rc = eprecon_kernel_map_async(d->table1, d->table_capacity, d->uniq1, d->m1, 3, 1, d->nbr1, stream);
    if (rc != EPRECON_OK) return rc;
    if (d->m2 > 0) rc = eprecon_kernel_map_async(d->table2, d->table_capacity, d->uniq2, d->m2, 3, 1, d->nbr2, stream);
    if (rc != EPRECON_OK) return rc;
    rc = eprecon_trilinear_map_async(d->table1, d->table_capacity, d->scaled1, d->n, 1, d->idx8_1, d->weight8_1, stream);
    if (rc != EPRECON_OK) return rc;
    if (!d->literal) return eprecon_trilinear_map_async(d->table2, d->table_capacity, d->scaled2, d->n, 1, d->idx8_2, d->weight8_2, stream);
    rc = eprecon_sphash_order_async(d->uniq1, d->m1, d->perm1, d->rank1, d->workspace, d->workspace_bytes, stream);
    if (rc != EPRECON_OK) return rc;
    rc = eprecon_sphash_order_async(d->uniq2, d->m2, d->perm2, d->rank2, d->workspace, d->workspace_bytes, stream);
    if (rc != EPRECON_OK) return rc;
    return eprecon_remap_index_async(d->idx8_1, d->n * 8, d->rank1, d->perm2, d->m2, d->idx8_2, stream);
}

}  // extern "C"
