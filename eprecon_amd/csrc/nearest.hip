// Nearest coarser-level voxel of every finest-level voxel on gfx950 (K18).
//
// Replaces torch.cdist + argmin at models/mask3dformer.py:361-367 of the reference ([N2 x N0] and
// [N2 x N1] fp32 distance matrices).  Coordinates are integers, so the nearest neighbour is decided
// on exact squared distances; ties go to the smallest row, which is what argmin returns.
// Search: the coarse voxels live on the grid of multiples of q.  With a = floor(c / q) * q the
// ancestor cell, any coarse voxel at least as close as an existing neighbour cell lies in the
// 4 x 4 x 4 block of cells a + q * {-1,0,1,2}^3 (per axis |r - q d| <= distance bound, r in [0,q)),
// probed through the hash grid.  If none of the 64 cells is occupied (the ancestor was pruned and
// no neighbour survives) the thread falls back to a scan of the whole coarse set — rare, and exact.
#include "hashgrid.hpp"

namespace {
using namespace ep;

__device__ __forceinline__ int fdiv(int a, int q) { return (a >= 0) ? a / q : -((-a + q - 1) / q); }

__global__ __launch_bounds__(256) void nearest_voxel_kernel(HashTable t, const int4 *coarse, int m,
                                                            const int4 *fine, int n, int q, int32_t *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = fine[i];
    const int ax = fdiv(c.y, q) * q, ay = fdiv(c.z, q) * q, az = fdiv(c.w, q) * q;
    long long best_d = 0x7fffffffffffffffll;
    int best = -1;
    for (int dx = -1; dx <= 2; ++dx)
        for (int dy = -1; dy <= 2; ++dy)
            for (int dz = -1; dz <= 2; ++dz) {
                const int x = ax + dx * q, y = ay + dy * q, z = az + dz * q;
                if (!key_in_range(c.x, x, y, z)) continue;
                const int j = hash_lookup(t, pack_key(c.x, x, y, z));
                if (j < 0) continue;
                const long long ex = c.y - x, ey = c.z - y, ez = c.w - z;
                const long long d = ex * ex + ey * ey + ez * ez;
                if (d < best_d || (d == best_d && j < best)) {
                    best_d = d;
                    best = j;
                }
            }
    // Every grid point outside the probed block differs from the query by at least 2q on some axis
    // (|r - q d| >= 2q for d <= -2 or d >= 3, r in [0,q)), so a block answer closer than 2q cannot
    // be beaten or tied from outside.  Otherwise (ancestor and all near neighbours absent): scan.
    if (best < 0 || best_d >= 4ll * q * q) {
        for (int j = 0; j < m; ++j) {
            const int4 o = coarse[j];
            if (o.x != c.x) continue;
            const long long ex = c.y - o.y, ey = c.z - o.z, ez = c.w - o.w;
            const long long d = ex * ex + ey * ey + ez * ez;
            if (d < best_d || (d == best_d && j < best)) {
                best_d = d;
                best = j;
            }
        }
    }
    out[i] = best;
}

}  // namespace

extern "C" int eprecon_nearest_voxel_async(const void *table, uint32_t capacity, const int32_t *coarse_coords,
                                           int64_t m, const int32_t *fine_coords, int64_t n, int quantum,
                                           int32_t *out_index, void *stream)
{
    if (!table || capacity < 1024 || (capacity & (capacity - 1)) || m < 0 || n < 0 || quantum < 1 ||
        (m > 0 && !coarse_coords) || (n > 0 && (!fine_coords || !out_index)))
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    HashTable t;
    t.keys = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(const_cast<void *>(table)) + 256);
    t.vals = reinterpret_cast<int32_t *>(t.keys + capacity);
    t.mask = capacity - 1;
    hipLaunchKernelGGL(nearest_voxel_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       t, reinterpret_cast<const int4 *>(coarse_coords), (int)m,
                       reinterpret_cast<const int4 *>(fine_coords), (int)n, quantum, out_index);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}
