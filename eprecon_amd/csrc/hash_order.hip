// The reference's voxel ORDER in one call: torchsparse numbers the voxels of initial_voxelize by ascending F.sphash
// (`torch.unique(pc_hash)`, ops/torchsparse_utils.py:19-21); ConvGRU's second gate convolution is the one consumer of that order
// (models/modules.py:216-217 with the stale corner indices of ops/torchsparse_utils.py:70-71,97-99; csrc/voxelize.hip,
// remap_stale_index_kernel).  Per voxel set: hash + iota -> rocPRIM radix sort of (hash, id) pairs over the hash's 60 bits ->
// perm[k] = id of the voxel with the k-th smallest hash, rank = its inverse.  One C call instead of a hash kernel, torch.sort
// (a merge sort: ~9 launches on 300k keys) and four index-glue ops, six times per fragment.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace {
using namespace ep;

__global__ void hash_iota_kernel(const int4 *coords_bxyz, int n, unsigned long long *hash, int32_t *iota)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = coords_bxyz[i];
    const int v[4] = {c.y, c.z, c.w, c.x};
    unsigned long long h = 14695981039346656037ull;     // FNV-1a over (x, y, z, batch), folded to 60 bits: csrc/voxelize.hip sphash_kernel
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h ^= (unsigned int)v[j];
        h *= 1099511628211ull;
    }
    hash[i] = (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFull);
    iota[i] = i;
}

__global__ void invert_perm_kernel(const int32_t *perm, int n, int32_t *rank)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) rank[perm[k]] = k;
}

size_t sort_temp_bytes(int n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0u, 60u, (hipStream_t)0);
    return bytes;
}

}  // namespace

extern "C" {

size_t eprecon_sphash_order_workspace_bytes(int64_t n)
{
    if (n <= 0 || n > 0x7fffffff) return 0;
    // [hash in | hash out | iota] + rocPRIM's scratch
    return align_up((size_t)n * 8, 256) * 2 + align_up((size_t)n * 4, 256) + align_up(sort_temp_bytes((int)n), 256);
}

int eprecon_sphash_order_async(const int32_t *coords, int64_t n, int32_t *perm_out, int32_t *rank_out, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    if (n < 0 || n > 0x7fffffff) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    if (!coords || !perm_out || !rank_out || !workspace) return EPRECON_ERR_ARG;
    if (workspace_bytes < eprecon_sphash_order_workspace_bytes(n)) return EPRECON_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)workspace;
    unsigned long long *h_in = (unsigned long long *)w;
    w += align_up((size_t)n * 8, 256);
    unsigned long long *h_out = (unsigned long long *)w;
    w += align_up((size_t)n * 8, 256);
    int32_t *iota = (int32_t *)w;
    w += align_up((size_t)n * 4, 256);
    size_t temp = sort_temp_bytes((int)n);
    const unsigned blocks = (unsigned)ceil_div(n, 256);
    hipLaunchKernelGGL(hash_iota_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const int4 *>(coords), (int)n, h_in, iota);
    EP_LAUNCH_CHECK();
    EP_HIP_CHECK(rocprim::radix_sort_pairs(w, temp, (const unsigned long long *)h_in, h_out, (const int32_t *)iota, perm_out, (size_t)n,
                                           0u, 60u, st));
    hipLaunchKernelGGL(invert_perm_kernel, dim3(blocks), dim3(256), 0, st, (const int32_t *)perm_out, (int)n, rank_out);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"
