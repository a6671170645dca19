// The rest of a GRU-fusion level's geometry in ONE call, after the level's host read has given the sizes
// (models/gru_fusion.py:332-347 -> models/modules.py:178-222 -> ops/torchsparse_utils.py:15-105): for the two voxelisations the
// six SConv3d of the level's two ConvGRUs share — point -> voxel CSR lists, 3x3x3 kernel maps, trilinear corner tables of the
// first, and for the reference-literal convr the hash order of both sets + the stale-index composition (csrc/hash_order.hip,
// voxelize.hip).  Nothing new is computed here: the call issues what the Python modules used to issue one by one (12 calls
// per level in a stretch of the fragment that is bound by the host's launch rate).
#include <mutex>
#include <vector>

#include "common.hpp"

namespace ep {
// The two voxelisations are independent chains of small launches (lists, kernel map, ten-launch radix sort each): the second
// chain is issued on a side stream of the library's own, forked from and joined to the caller's stream with events.  Inside
// one C call the launches are issued in microseconds, so the GPU — a chain of dependent 5 us kernels — is what the time goes
// to, and two chains side by side take about half of it.
static std::mutex g_fork_mutex;
static std::vector<Fork> g_forks;

int fork_for(hipStream_t main, Fork &out)
{
    std::lock_guard<std::mutex> lock(g_fork_mutex);
    for (const Fork &f : g_forks)
        if (f.main == main) {
            out = f;
            return EPRECON_OK;
        }
    Fork f;
    f.main = main;
    EP_HIP_CHECK(hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking));
    EP_HIP_CHECK(hipEventCreateWithFlags(&f.ev_fork, hipEventDisableTiming));
    EP_HIP_CHECK(hipEventCreateWithFlags(&f.ev_join, hipEventDisableTiming));
    g_forks.push_back(f);
    out = f;
    return EPRECON_OK;
}
}  // namespace ep
using ep::Fork;
using ep::fork_for;

extern "C" {

static size_t finish_half_bytes(int64_t n, int64_t m1, int64_t m2)
{
    size_t w = eprecon_segment_workspace_bytes(n, m1);
    const size_t cands[] = {eprecon_segment_workspace_bytes(n, m2), eprecon_sphash_order_workspace_bytes(m1),
                            eprecon_sphash_order_workspace_bytes(m2)};
    for (size_t c : cands) w = c > w ? c : w;
    return ep::align_up(w, 256);
}

size_t eprecon_gru_stage_finish_workspace_bytes(int64_t n, int64_t m1, int64_t m2) { return 2 * finish_half_bytes(n, m1, m2); }

int eprecon_gru_stage_finish_async(const eprecon_gru_finish_desc *d, void *stream)
{
    if (!d || d->n < 0 || d->m1 < 0 || d->m2 < 0) return EPRECON_ERR_ARG;
    if (d->n == 0) return EPRECON_OK;
    if (d->workspace_bytes < eprecon_gru_stage_finish_workspace_bytes(d->n, d->m1, d->m2)) return EPRECON_ERR_WORKSPACE;
    const size_t half = finish_half_bytes(d->n, d->m1, d->m2);
    void *ws_a = d->workspace, *ws_b = (char *)d->workspace + half;
    hipStream_t main = (hipStream_t)stream;
    Fork f;
    int rc = fork_for(main, f);
    if (rc != EPRECON_OK) return rc;
    void *side = (void *)f.side;
#define EP_STEP(call)                      \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != EPRECON_OK) return rc_; \
    } while (0)
    // Each chain is a lambda so that a failing step cannot skip the join: whatever was issued on the side stream is
    // ordered before the caller's later work either way, and the first error is what the call returns.
    auto second = [&]() -> int {  // ---- second voxelisation: side stream ----
        EP_STEP(eprecon_segment_lists_async(d->inverse2, d->n, d->m2, d->offsets2, d->order2, ws_b, half, side));
        if (d->m2 > 0) EP_STEP(ep::kernel_map_self_prefilled(d->table2, d->table_capacity, d->uniq2, d->m2, 1, d->nbr2, side));
        if (d->literal)
            EP_STEP(eprecon_sphash_order_async(d->uniq2, d->m2, d->perm2, d->rank2, ws_b, half, side));
        else
            EP_STEP(d->m2 > 0 ? ep::trilinear_from_map(d->scaled2, d->n, d->inverse2, d->nbr2, d->m2, 1, d->idx8_2, d->weight8_2, side)
                              : eprecon_trilinear_map_async(d->table2, d->table_capacity, d->scaled2, d->n, 1, d->idx8_2, d->weight8_2, side));
        return EPRECON_OK;
    };
    auto first = [&]() -> int {  // ---- first voxelisation: the caller's stream ----
        EP_STEP(eprecon_segment_lists_async(d->inverse1, d->n, d->m1, d->offsets1, d->order1, ws_a, half, stream));
        if (d->m1 > 0) EP_STEP(ep::kernel_map_self_prefilled(d->table1, d->table_capacity, d->uniq1, d->m1, 1, d->nbr1, stream));
        // (corners from the set's kernel map and the numbering's inverse: no hash probes, ep::trilinear_from_map)
        EP_STEP(d->m1 > 0 ? ep::trilinear_from_map(d->scaled1, d->n, d->inverse1, d->nbr1, d->m1, 1, d->idx8_1, d->weight8_1, stream)
                          : eprecon_trilinear_map_async(d->table1, d->table_capacity, d->scaled1, d->n, 1, d->idx8_1, d->weight8_1, stream));
        if (d->literal) EP_STEP(eprecon_sphash_order_async(d->uniq1, d->m1, d->perm1, d->rank1, ws_a, half, stream));
        return EPRECON_OK;
    };
    {   // the upper halves of the two self maps hold -1 before the mirrored entries are scattered into them: ONE launch, before the fork
        ep::FillRegion reg[2] = {ep::kernel_map_self_fill_region(d->nbr1, d->m1), ep::kernel_map_self_fill_region(d->nbr2, d->m2)};
        rc = ep::multi_fill(reg, 2, main);
        if (rc != EPRECON_OK) return rc;
    }
    EP_HIP_CHECK(hipEventRecord(f.ev_fork, main));
    EP_HIP_CHECK(hipStreamWaitEvent(f.side, f.ev_fork, 0));
    const int rc_second = second();
    EP_HIP_CHECK(hipEventRecord(f.ev_join, f.side));
    const int rc_first = first();
    EP_HIP_CHECK(hipStreamWaitEvent(main, f.ev_join, 0));
    if (rc_second != EPRECON_OK) return rc_second;
    if (rc_first != EPRECON_OK) return rc_first;
    if (d->literal) EP_STEP(eprecon_remap_index_async(d->idx8_1, d->n * 8, d->rank1, d->perm2, d->m2, d->idx8_2, stream));
#undef EP_STEP
    return EPRECON_OK;
}

}  // extern "C"
