// The reference's voxel ORDER in one call: torchsparse numbers the voxels of initial_voxelize by ascending F.sphash
// (`torch.unique(pc_hash)`, ops/torchsparse_utils.py:19-21); ConvGRU's second gate convolution is the one consumer of that order
// (models/modules.py:216-217 with the stale corner indices of ops/torchsparse_utils.py:70-71,97-99; csrc/voxelize.hip,
// remap_stale_index_kernel).  Per voxel set: perm[k] = id of the voxel with the k-th smallest 60-bit hash, rank = its inverse.
//
// Round 5: a bucket sort of its own instead of rocPRIM's radix sort of (hash, id) pairs.  The library sort is a merge sort at
// these sizes (10k-330k keys): 8-10 launches of ~6 us per call, six calls per fragment = the largest single item of the
// GRU-fusion bookkeeping (0.47 of its 1.9 ms of kernels, profiles/r04/cfg4_layers.txt).  The hashes are FNV-1a values, i.e.
// uniform: the top B bits (B chosen so that a bucket holds ~8 keys) split the set evenly, and inside a bucket every key finds its
// place by COUNTING the bucket's smaller (hash, id) pairs — no data-dependent sorting network, no ordering left to atomics:
//   memset | hash + bucket histogram | exclusive scan (one launch, <= 32768 buckets) | scatter into the buckets (arbitrary order
//   inside one) | rank inside the bucket -> perm, rank                                                        = 5 launches.
// The result is the sorted order itself ((hash, id) ascending: what a stable sort of the hashes gives), so it is bit-identical
// to the library sort's and to the oracle's (tests/test_spvcnn_gpu.py::test_sphash_order_matches_oracle).
#include <cstring>
#include <string.h>

#include "common.hpp"

namespace {
using namespace ep;

__device__ __forceinline__ unsigned long long sphash60(const int4 c)
{
    const int v[4] = {c.y, c.z, c.w, c.x};
    unsigned long long h = 14695981039346656037ull;     // FNV-1a over (x, y, z, batch), folded to 60 bits: csrc/voxelize.hip sphash_kernel
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h ^= (unsigned int)v[j];
        h *= 1099511628211ull;
    }
    return (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFull);
}

__global__ void hash_count_kernel(const int4 *coords_bxyz, int n, int shift, unsigned long long *hash, int32_t *count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long h = sphash60(coords_bxyz[i]);
    hash[i] = h;
    atomicAdd(&count[(int)(h >> shift)], 1);      // (integer counts: the same whatever the order)
}

// slot order inside a bucket is whatever the atomics give; nothing downstream depends on it (bucket_rank_kernel counts)
__global__ void hash_scatter_kernel(const unsigned long long *hash, int n, int shift, const int32_t *start, int32_t *cursor,
                                    int32_t *slot_id)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(hash[i] >> shift);
    slot_id[start[b] + atomicAdd(&cursor[b], 1)] = i;
}

// one thread per slot: the final position of its voxel = bucket start + number of (hash, id) pairs of the bucket below its own
__global__ void bucket_rank_kernel(const unsigned long long *hash, const int32_t *slot_id, int n, int shift, const int32_t *start,
                                   const int32_t *count, int32_t *perm, int32_t *rank)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int id = slot_id[s];
    const unsigned long long h = hash[id];
    const int b = (int)(h >> shift);
    const int lo = start[b], cnt = count[b];
    int below = 0;
    for (int t = lo; t < lo + cnt; ++t) {
        const int other = slot_id[t];
        const unsigned long long ho = hash[other];
        below += (ho < h || (ho == h && other < id)) ? 1 : 0;
    }
    perm[lo + below] = id;
    rank[id] = lo + below;
}

// buckets = 2^bits with ~8 keys each, at most 32768 (the one-launch scan), at least 16
int bucket_bits(int64_t n)
{
    int bits = 4;
    while (bits < 15 && ((int64_t)8 << bits) < n) ++bits;
    return bits;
}

}  // namespace

extern "C" {

size_t eprecon_sphash_order_workspace_bytes(int64_t n)
{
    if (n <= 0 || n > 0x7fffffff) return 0;
    const size_t nb = (size_t)1 << bucket_bits(n);
    // [hash | slot ids | count, cursor | start | scan scratch]
    return align_up((size_t)n * 8, 256) + align_up((size_t)n * 4, 256) + 3 * align_up(nb * 4, 256) + 512;
}

int eprecon_sphash_order_async(const int32_t *coords, int64_t n, int32_t *perm_out, int32_t *rank_out, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    if (n < 0 || n > 0x7fffffff) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    if (!coords || !perm_out || !rank_out || !workspace) return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_sphash_order_workspace_bytes(n)) return EPRECON_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int bits = bucket_bits(n), shift = 60 - bits;
    const size_t nb = (size_t)1 << bits, seg = align_up(nb * 4, 256);
    char *w = (char *)workspace;
    unsigned long long *hash = (unsigned long long *)w;
    w += align_up((size_t)n * 8, 256);
    int32_t *slot_id = (int32_t *)w;
    w += align_up((size_t)n * 4, 256);
    int32_t *count = (int32_t *)w, *cursor = (int32_t *)(w + seg), *start = (int32_t *)(w + 2 * seg);
    int32_t *scratch = (int32_t *)(w + 3 * seg);
    const unsigned blocks = (unsigned)ceil_div(n, 256);
    const int4 *c4 = reinterpret_cast<const int4 *>(coords);
    EP_HIP_CHECK(hipMemsetAsync(count, 0, 2 * seg, st));       // count and cursor
    hipLaunchKernelGGL(hash_count_kernel, dim3(blocks), dim3(256), 0, st, c4, (int)n, shift, hash, count);
    EP_LAUNCH_CHECK();
    const int rc = ep::exclusive_scan_i32(count, (int)nb, start, scratch, nullptr, st);
    if (rc != EPRECON_OK) return rc;
    hipLaunchKernelGGL(hash_scatter_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned long long *)hash, (int)n, shift,
                       (const int32_t *)start, cursor, slot_id);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bucket_rank_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned long long *)hash,
                       (const int32_t *)slot_id, (int)n, shift, (const int32_t *)start, (const int32_t *)count, perm_out, rank_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"
