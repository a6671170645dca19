// Coordinate bookkeeping for the sparse layers on gfx950: hash-grid build / query, unique
// voxel ids (stable, first-occurrence order), stride-1 and strided kernel maps.
//
// Replaces (reference call sites; implementations are in the un-vendored torchsparse / spconv):
//   F.sphash + F.sphashquery            ops/torchsparse_utils.py:19-21,44-50,73-79
//   torch.unique(hash) voxel ids        ops/torchsparse_utils.py:20-22   (initial_voxelize)
//   spnn.Conv3d kernel maps (k3 s1, k2 s2 and its transpose), spconv SubMConv3d indice pairs
//                                       models/modules.py:19-64,90-122,181,252,444
// Ordering contract of this build (the reference's voxel order is "ascending hash", i.e. arbitrary
// and never relied on — SURVEY.md appendix A.2): unique voxels are numbered in order of FIRST
// OCCURRENCE in the input list.  All kernels are atomic-free in their outputs except the hash
// insert (atomicCAS on the key, atomicMin on the row index), whose result is order-independent.
#include <stdlib.h>

#include <atomic>

#include "hashgrid.hpp"

namespace {
using namespace ep;

// -------------------------------------------------------------------------------------------
// device-wide exclusive scan of int32 (n up to 2^31), one to three launches, deterministic
// -------------------------------------------------------------------------------------------
constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;                       // per thread
constexpr int kScanTile = kScanBlock * kScanItems;  // 2048 per block

// (n_dev, optional: the live length is min(n, *n_dev) — the launch is sized by the capacity n, the count lives on the device)
__global__ __launch_bounds__(kScanBlock) void scan_tile_sums(const int32_t *in, int n, int32_t *sums, const int32_t *n_dev)
{
    __shared__ int sWave[kScanBlock / kWave];
    if (n_dev) n = min(n, *n_dev);
    const int base = blockIdx.x * kScanTile;
    int s = 0;
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k * kScanBlock + threadIdx.x;
        if (i < n) s += in[i];
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & (kWave - 1)) == 0) sWave[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kScanBlock / kWave; ++w) t += sWave[w];
        sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void scan_sums_inplace(int32_t *sums, int nblk, int32_t *total)
{
    __shared__ int sWave[1024 / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    int carry = 0;
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + tid;
        const int v = i < nblk ? sums[i] : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == kWave - 1) sWave[wid] = x;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 1024 / kWave; ++w) {
            const int c = sWave[w];
            woff += (w < wid) ? c : 0;
            tot += c;
        }
        if (i < nblk) sums[i] = carry + woff + x - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0 && total) *total = carry;
}

// out[i] = exclusive prefix of in[0..i); each block rescans its 2048-element tile in LDS.
// RAW: `sums` holds the tile totals as scan_tile_sums left them and every workgroup adds up the totals in front of its own
// tile itself (a few hundred values: cheaper than the third launch that used to scan them); the last workgroup also
// publishes the grand total.
template <bool RAW>
__global__ __launch_bounds__(kScanBlock) void scan_apply(const int32_t *in, int n, const int32_t *sums,
                                                         int32_t *out, const int32_t *n_dev, int32_t *total)
{
    __shared__ int sWave[kScanBlock / kWave];
    __shared__ int sBase;
    if (n_dev) n = min(n, *n_dev);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    int tile_base;
    if (RAW) {
        int acc = 0;
        for (int j = tid; j < (int)blockIdx.x; j += kScanBlock) acc += sums[j];
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) sWave[wid] = acc;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < kScanBlock / kWave; ++w) t += sWave[w];
            sBase = t;
            if (total && blockIdx.x == gridDim.x - 1) *total = t + sums[blockIdx.x];
        }
        __syncthreads();
        tile_base = sBase;
        __syncthreads();      // (sWave is reused below)
    } else {
        tile_base = sums[blockIdx.x];
    }
    const int base = blockIdx.x * kScanTile + tid * kScanItems;  // blocked arrangement
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int x = s;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == kWave - 1) sWave[wid] = x;
    __syncthreads();
    int off = tile_base + x - s;
    for (int w = 0; w < wid; ++w) off += sWave[w];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = off;
        off += v[k];
    }
}

// short inputs: the whole scan in ONE workgroup and (round 6) ONE pass: 1024 threads x ITEMS items held in registers (8 / 16 /
// 32 by length), one barrier — the carried rounds of the first form (8 items per thread and round, two barriers each) made a
// 32,768-element scan a chain of four round trips: 12 us per launch, 21 launches per cfg4 fragment
constexpr int kSmallScanMax = 32768;
template <int ITEMS>
__global__ __launch_bounds__(1024) void scan_small_kernel(const int32_t *in, int n, int32_t *out, int32_t *total,
                                                          const int32_t *n_dev)
{
    __shared__ int sWave[1024 / kWave];
    if (n_dev) n = min(n, *n_dev);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    const int b0 = tid * ITEMS;  // blocked arrangement
    int v[ITEMS];
    int s = 0;
    if (b0 + ITEMS <= n && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < ITEMS; k += 4) {
            const int4 q = *reinterpret_cast<const int4 *>(in + b0 + k);
            v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) v[k] = (b0 + k < n) ? in[b0 + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) s += v[k];
    int x = s;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == kWave - 1) sWave[wid] = x;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 1024 / kWave; ++w) {
        const int c = sWave[w];
        woff += (w < wid) ? c : 0;
        tot += c;
    }
    int off = woff + x - s;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if (b0 + k < n) out[b0 + k] = off;
        off += v[k];
    }
    if (tid == 0 && total) *total = tot;
}

// ---- long inputs in ONE launch (round 6): decoupled look-back -------------------------------------------------------------
// A tile (256 threads x 8 items) takes its index from a device counter (tiles start in index order, so every predecessor of a
// running tile has started: the look-back cannot wait for a workgroup that is not scheduled), publishes its total as
// {1, sum}, sums its predecessors' words 64 at a time with one wave until it meets an inclusive prefix {2, ...}, publishes its
// own inclusive prefix and scans its items behind that base.  One 64-bit word per tile carries flag and value together (no
// fence: a single agent-scope atomic store / load).  The words and the two counters live in a zero-initialised device array, one
// set per launch in rotation; the last tile to finish clears its set (graph replays reuse theirs).  Replaces tile-sums + apply
// (two launches) for up to kLbMaxTiles tiles; bit-identical results (integer sums).
constexpr int kLbSlots = 64, kLbMaxTiles = 4096;
__device__ unsigned long long g_lb_state[kLbSlots * kLbMaxTiles];
__device__ unsigned int g_lb_ctr[kLbSlots * 2];

__global__ __launch_bounds__(kScanBlock) void scan_lookback_kernel(const int32_t *in, int n, int32_t *out, int32_t *total,
                                                                   const int32_t *n_dev, unsigned long long *state, unsigned int *ctr)
{
    __shared__ int sWave[kScanBlock / kWave];
    __shared__ int sTile, sBase;
    if (n_dev) n = min(n, *n_dev);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    const int ntiles = (int)gridDim.x;
    if (tid == 0) sTile = (int)atomicAdd(&ctr[0], 1u);
    __syncthreads();
    const int tile = sTile;
    const int base = tile * kScanTile + tid * kScanItems;  // blocked arrangement
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int x = s;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == kWave - 1) sWave[wid] = x;
    __syncthreads();
    int woff = 0, agg = 0;
#pragma unroll
    for (int w = 0; w < kScanBlock / kWave; ++w) {
        const int c = sWave[w];
        woff += (w < wid) ? c : 0;
        agg += c;
    }
    if (wid == 0) {      // one wave publishes and looks back
        int prefix = 0;
        if (tile > 0) {
            if (lane == 0)
                __hip_atomic_store(&state[tile], (1ull << 62) | (unsigned int)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int win = tile - 1;           // (wave-uniform) nearest predecessor of the current window of 64
            while (true) {
                const int j = win - lane;
                unsigned long long w = 0ull;
                if (j >= 0) {
                    do {
                        w = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((w >> 62) == 0ull);
                }
                // lanes are ordered nearest predecessor first: everything up to (and including) the first inclusive prefix counts
                const unsigned long long incl = __ballot(j >= 0 && (w >> 62) == 2ull);
                const int first = incl ? __builtin_ctzll(incl) : kWave;
                int part = (j >= 0 && lane <= first) ? (int)(unsigned int)w : 0;
#pragma unroll
                for (int d = kWave / 2; d > 0; d >>= 1) part += __shfl_xor(part, d);
                prefix += part;
                if (incl || win < kWave) break;      // an inclusive prefix closes the sum; so does reaching tile 0
                win -= kWave;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&state[tile], (2ull << 62) | (unsigned int)(prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sBase = prefix;
            if (total && tile == ntiles - 1) *total = prefix + agg;
        }
    }
    __syncthreads();
    int off = sBase + woff + x - s;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = off;
        off += v[k];
    }
    // the last tile to get here puts the set back to zero (every tile has read what it needed: it is past its look-back)
    __syncthreads();
    if (tid == 0) sTile = (int)atomicAdd(&ctr[1], 1u);
    __syncthreads();
    if (sTile == ntiles - 1) {
        for (int t = tid; t < ntiles; t += kScanBlock) __hip_atomic_store(&state[t], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(&ctr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctr[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

// EPRECON_SCAN_LOOKBACK=0: long scans as tile sums + apply (two launches, rounds 1-5; read per call)
static bool lookback_enabled()
{
    const char *e = getenv("EPRECON_SCAN_LOOKBACK");
    return !(e && e[0] == '0');
}

namespace ep {
// scratch: ceil(n / 2048) int32.  `total_dev` (optional) receives the grand total.  n_dev (optional, device): the live
// length is min(n, *n_dev); launches are sized by n.
int exclusive_scan_i32_dn(const int32_t *in, int n, const int32_t *n_dev, int32_t *out, int32_t *scratch, int32_t *total_dev,
                          hipStream_t st)
{
    if (n <= 0) {
        if (total_dev) EP_HIP_CHECK(hipMemsetAsync(total_dev, 0, sizeof(int32_t), st));
        return EPRECON_OK;
    }
    if (n <= kSmallScanMax) {
        if (n <= 8192) hipLaunchKernelGGL(scan_small_kernel<8>, dim3(1), dim3(1024), 0, st, in, n, out, total_dev, n_dev);
        else if (n <= 16384) hipLaunchKernelGGL(scan_small_kernel<16>, dim3(1), dim3(1024), 0, st, in, n, out, total_dev, n_dev);
        else hipLaunchKernelGGL(scan_small_kernel<32>, dim3(1), dim3(1024), 0, st, in, n, out, total_dev, n_dev);
        EP_LAUNCH_CHECK();
        return EPRECON_OK;
    }
    const int nblk = (int)ceil_div(n, kScanTile);
    if (nblk <= kLbMaxTiles && lookback_enabled()) {
        static unsigned long long *state_base = nullptr;
        static unsigned int *ctr_base = nullptr;
        if (!state_base) {
            EP_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void **>(&state_base), HIP_SYMBOL(g_lb_state)));
            EP_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void **>(&ctr_base), HIP_SYMBOL(g_lb_ctr)));
        }
        static std::atomic<unsigned> slot{0};
        const unsigned sidx = slot.fetch_add(1u) % (unsigned)kLbSlots;
        hipLaunchKernelGGL(scan_lookback_kernel, dim3(nblk), dim3(kScanBlock), 0, st, in, n, out, total_dev, n_dev,
                           state_base + (size_t)sidx * kLbMaxTiles, ctr_base + 2 * sidx);
        EP_LAUNCH_CHECK();
        return EPRECON_OK;
    }
    hipLaunchKernelGGL(scan_tile_sums, dim3(nblk), dim3(kScanBlock), 0, st, in, n, scratch, n_dev);
    EP_LAUNCH_CHECK();
    if (nblk <= 4096) {       // (8.4 M elements: two launches; the tile totals are summed by the consumers)
        hipLaunchKernelGGL(scan_apply<true>, dim3(nblk), dim3(kScanBlock), 0, st, in, n, scratch, out, n_dev, total_dev);
        EP_LAUNCH_CHECK();
        return EPRECON_OK;
    }
    hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(1024), 0, st, scratch, nblk, total_dev);
    EP_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_apply<false>, dim3(nblk), dim3(kScanBlock), 0, st, in, n, scratch, out, n_dev, (int32_t *)nullptr);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}
int exclusive_scan_i32(const int32_t *in, int n, int32_t *out, int32_t *scratch, int32_t *total_dev, hipStream_t st)
{
    return exclusive_scan_i32_dn(in, n, nullptr, out, scratch, total_dev, st);
}
}  // namespace ep

namespace {

__device__ __forceinline__ int floor_div(int a, int q) { return (a >= 0) ? a / q : -((-a + q - 1) / q); }

__global__ void hash_clear_kernel(unsigned long long *keys, int32_t *vals, uint32_t cap, int32_t *status)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        keys[i] = kEmptyKey;
        vals[i] = 0x7fffffff;
    }
    if (i == 0) status[0] = 0;
}

// coords int32[n,4] (b,x,y,z); key = floor(c / q) * q per spatial coordinate (q >= 1)
__global__ void hash_insert_kernel(HashTable t, const int4 *coords, int n, int q, int32_t *status, const int32_t *n_dev)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (i >= n) return;
    int4 c = coords[i];
    if (q > 1) {
        c.y = floor_div(c.y, q) * q;
        c.z = floor_div(c.z, q) * q;
        c.w = floor_div(c.w, q) * q;
    }
    if (!key_in_range(c.x, c.y, c.z, c.w)) {
        atomicOr(status, 1);
        return;
    }
    if (!hash_insert(t, pack_key(c.x, c.y, c.z, c.w), i)) atomicOr(status, 2);
}

__global__ void hash_query_kernel(HashTable t, const int4 *queries, int m, int q, int32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int4 c = queries[i];
    if (q > 1) {
        c.y = floor_div(c.y, q) * q;
        c.z = floor_div(c.z, q) * q;
        c.w = floor_div(c.w, q) * q;
    }
    out[i] = key_in_range(c.x, c.y, c.z, c.w) ? hash_lookup(t, pack_key(c.x, c.y, c.z, c.w)) : -1;
}

// first[i] = 1 when row i is the first occurrence of its (quantised) key
__global__ void first_flag_kernel(HashTable t, const int4 *coords, int n, int q, int32_t *first,
                                  int32_t *owner, const int32_t *n_dev)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (i >= n) return;
    int4 c = coords[i];
    if (q > 1) {
        c.y = floor_div(c.y, q) * q;
        c.z = floor_div(c.z, q) * q;
        c.w = floor_div(c.w, q) * q;
    }
    const int o = key_in_range(c.x, c.y, c.z, c.w) ? hash_lookup(t, pack_key(c.x, c.y, c.z, c.w)) : -1;
    owner[i] = o;
    first[i] = (o == i) ? 1 : 0;
}

// inverse[i] = unique id of row i; unique_coords[uid] = quantised coords of the first occurrence;
// the table values are rewritten from "first row" to "unique id" so later lookups return ids.
// (the launch also covers the table: thread s < cap rewrites slot s — table_vals_to_ids_kernel's job; the two halves touch
// disjoint data, so one launch serves both)
__global__ void unique_finalize_kernel(HashTable t, const int4 *coords, int n, int q,
                                       const int32_t *owner, const int32_t *rank, int32_t *inverse,
                                       int4 *unique_coords, const int32_t *n_dev, uint32_t cap, const int32_t *status,
                                       int32_t *status_copy)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && status_copy) *status_copy = *status;   // (the inserts finished a launch ago: the word is final)
    if ((uint32_t)i < cap && t.keys[i] != kEmptyKey) t.vals[i] = rank[t.vals[i]];
    if (n_dev) n = min(n, *n_dev);
    if (i >= n) return;
    const int o = owner[i];
    if (o < 0) {
        inverse[i] = -1;
        return;
    }
    const int uid = rank[o];
    inverse[i] = uid;
    if (o == i) {
        int4 c = coords[i];
        if (q > 1) {
            c.y = floor_div(c.y, q) * q;
            c.z = floor_div(c.z, q) * q;
            c.w = floor_div(c.w, q) * q;
        }
        unique_coords[uid] = c;
    }
}

// nbr[k][i] = row of the voxel at coords[i] + offset_k * stride (or -1).
// ksize 3: 27 offsets, x fastest (k = ((dz+1)*3 + (dy+1))*3 + (dx+1));
// ksize 2: 8 offsets in {0,1}^3, z fastest (k = 4*bx + 2*by + bz)   
__global__ void kernel_map_kernel(HashTable t, const int4 *coords, int n, int ksize, int stride,
                                  int32_t *nbr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= n) return;
    const int4 c = coords[i];
    int dx, dy, dz;
    if (ksize == 3) {
        dx = k % 3 - 1;
        dy = (k / 3) % 3 - 1;
        dz = k / 9 - 1;
    } else {
        dx = (k >> 2) & 1;
        dy = (k >> 1) & 1;
        dz = k & 1;
    }
    const int x = c.y + dx * stride, y = c.z + dy * stride, z = c.w + dz * stride;
    nbr[(size_t)k * n + i] = key_in_range(c.x, x, y, z) ? hash_lookup(t, pack_key(c.x, x, y, z)) : -1;
}

// The same 3x3x3 map for a voxel set queried against ITS OWN table (row i of `coords` is the table's value for its key: the sets
// the unique numbering produces): offset -o from voxel j leads to voxel i exactly when offset +o from i leads to j, so only the 13
// offsets below the centre (and the centre) are looked up and every hit also fills the mirrored entry nbr[26 - k][j] = i.  The
// upper half is pre-filled with -1 by the caller (hipMemsetAsync); a kernel map is injective per offset, so no two threads write
// the same mirrored entry.  Half the hash probes of kernel_map_kernel (the probes, random reads of a table of tens of MB, are
// what that kernel costs: 0.6 ms per cfg4 fragment), the same table bit for bit.
__global__ void kernel_map_self_kernel(HashTable t, const int4 *coords, int n, int stride, int32_t *nbr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;      // 0 .. 13
    if (i >= n) return;
    const int4 c = coords[i];
    const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
    const int x = c.y + dx * stride, y = c.z + dy * stride, z = c.w + dz * stride;
    const int j = key_in_range(c.x, x, y, z) ? hash_lookup(t, pack_key(c.x, x, y, z)) : -1;
    nbr[(size_t)k * n + i] = j;
    if (k < 13 && j >= 0) nbr[(size_t)(26 - k) * n + j] = i;
}

// transposed k2s2 map: up[k][i] = parent row of fine voxel i when i is child k of its parent, else -1
__global__ void transpose_map_kernel(const int4 *fine, int n, const int32_t *parent, int fine_stride,
                                     int32_t *up)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = fine[i];
    const int q = 2 * fine_stride;
    const int bx = (c.y - floor_div(c.y, q) * q) / fine_stride;
    const int by = (c.z - floor_div(c.z, q) * q) / fine_stride;
    const int bz = (c.w - floor_div(c.w, q) * q) / fine_stride;
    const int kk = 4 * bx + 2 * by + bz;
    const int par = parent[i];
    for (int k = 0; k < 8; ++k) up[(size_t)k * n + i] = (k == kk) ? par : -1;
}

// rank volume of a voxel set on a dense grid: rank[(x/stride * gy + y/stride) * gz + z/stride] = row (cells start at -1)
__global__ void grid_rank_kernel(const int4 *coords, int n, int stride, int gx, int gy, int gz, int32_t *rank, int32_t *bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = coords[i];
    const int x = c.y / stride, y = c.z / stride, z = c.w / stride;
    const bool ok = c.y >= 0 && c.z >= 0 && c.w >= 0 && x < gx && y < gy && z < gz && x * stride == c.y && y * stride == c.z &&
                    z * stride == c.w;
    if (ok) rank[((size_t)x * gy + y) * gz + z] = i;
    else atomicAdd(bad, 1);
}

HashTable make_table(void *mem, uint32_t cap)
{
    HashTable t;
    t.keys = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(mem) + 256);
    t.vals = reinterpret_cast<int32_t *>(t.keys + cap);
    t.mask = cap - 1;
    return t;
}
int32_t *table_status(void *mem) { return reinterpret_cast<int32_t *>(mem); }
bool is_pow2(uint32_t c) { return c >= 1024 && (c & (c - 1)) == 0; }


// Kernel map of a dense 2D 'same' convolution over `maps` images of h x w pixels stored row-major
// ([maps][h][w] rows of a channels-last tensor): nbr[(ky * ks + kx)][i] = row of pixel
// (y + ky - ks/2, x + kx - ks/2) of the same image, -1 outside it (zero padding).  Closed form, no
// hash grid: lets the 2D fusion convolutions of Occupancy_Initialization
// (models/occupancy_initialization.py:22-31, models/modules.py:313-399) run on the gather-GEMM kernel.
__global__ __launch_bounds__(256) void pixel_map_kernel(int maps, int h, int w, int ks, int32_t *nbr)
{
    const int n = maps * h * w;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = blockIdx.y;
    const int dy = k / ks - ks / 2, dx = k % ks - ks / 2;
    const int x = i % w, y = (i / w) % h;
    const int yy = y + dy, xx = x + dx;
    nbr[(size_t)k * n + i] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? i + dy * w + dx : -1;
}

}  // namespace

namespace ep {
namespace {
struct FillParams {
    uint32_t *base[kMaxFillRegions];            // the region's start rounded DOWN to 16 bytes
    unsigned long long end[kMaxFillRegions];    // exclusive prefix end, in 16-byte chunks
    unsigned lead[kMaxFillRegions];             // 32-bit words of the first chunk in front of the region
    unsigned long long words[kMaxFillRegions];  // the region's length in 32-bit words
    uint32_t value[kMaxFillRegions];
    int count;
};
// one thread per 16-byte chunk; chunks that straddle a region's first / last word write word by word
__global__ __launch_bounds__(256) void multi_fill_kernel(FillParams f)
{
    const unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    unsigned long long begin = 0;
#pragma unroll 1
    for (int r = 0; r < f.count; ++r) {
        if (e < f.end[r]) {
            const uint32_t v = f.value[r];
            const unsigned long long w0 = (e - begin) * 4;                    // first word of the chunk, from the rounded-down base
            const unsigned long long lo = f.lead[r], hi = f.lead[r] + f.words[r];
            uint32_t *p = f.base[r] + w0;
            if (w0 >= lo && w0 + 4 <= hi) {
                *reinterpret_cast<uint4 *>(p) = make_uint4(v, v, v, v);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (w0 + k >= lo && w0 + k < hi) p[k] = v;
            }
            return;
        }
        begin = f.end[r];
    }
}
}  // namespace

int multi_fill(const FillRegion *regions, int count, hipStream_t st)
{
    if (count < 0 || count > kMaxFillRegions) return EPRECON_ERR_ARG;
    FillParams f;
    f.count = 0;
    unsigned long long total = 0;
    for (int r = 0; r < count; ++r) {
        if (regions[r].bytes == 0) continue;
        const uintptr_t a = reinterpret_cast<uintptr_t>(regions[r].p);
        if (!regions[r].p || (a & 3) || (regions[r].bytes & 3)) return EPRECON_ERR_ARG;
        const unsigned lead = (unsigned)((a & 15) / 4);
        const unsigned long long words = regions[r].bytes / 4;
        total += (lead + words + 3) / 4;
        f.base[f.count] = reinterpret_cast<uint32_t *>(a & ~(uintptr_t)15);
        f.lead[f.count] = lead;
        f.words[f.count] = words;
        f.end[f.count] = total;
        f.value[f.count] = regions[r].value;
        ++f.count;
    }
    if (total == 0) return EPRECON_OK;
    hipLaunchKernelGGL(multi_fill_kernel, dim3((unsigned)ceil_div((int64_t)total, 256)), dim3(256), 0, st, f);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int table_clear_regions(void *table, uint32_t capacity, FillRegion *out3)
{
    if (!table || !is_pow2(capacity)) return EPRECON_ERR_ARG;
    HashTable t = make_table(table, capacity);
    out3[0] = FillRegion{table, 16, 0u};                                          // status word (and the unique count next to it)
    out3[1] = FillRegion{t.keys, (size_t)capacity * 8, 0xFFFFFFFFu};              // kEmptyKey
    out3[2] = FillRegion{t.vals, (size_t)capacity * 4, 0x7fffffffu};
    return EPRECON_OK;
}
}  // namespace ep

extern "C" {


uint32_t eprecon_hash_capacity(int64_t n) { return ep::hash_capacity_for(n); }
size_t eprecon_hash_table_bytes(uint32_t capacity) { return 256 + (size_t)capacity * 12; }

static int hash_build_impl(const int32_t *coords, int64_t n, const int32_t *n_dev, int quantum, void *table, uint32_t capacity,
                           void *stream, bool cleared = false)
{
    if (!table || !is_pow2(capacity) || n < 0 || quantum < 1 || (uint64_t)capacity < (uint64_t)n + 1 ||
        (n > 0 && !coords))
        return EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HashTable t = make_table(table, capacity);
    if (!cleared) {   // (cleared: the caller reset the table with its other regions, ep::multi_fill + ep::table_clear_regions)
        hipLaunchKernelGGL(hash_clear_kernel, dim3((capacity + 255) / 256), dim3(256), 0, st, t.keys, t.vals,
                           capacity, table_status(table));
        EP_LAUNCH_CHECK();
    }
    if (n > 0) {
        hipLaunchKernelGGL(hash_insert_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, t,
                           reinterpret_cast<const int4 *>(coords), (int)n, quantum, table_status(table), n_dev);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}

int eprecon_hash_build_async(const int32_t *coords, int64_t n, int quantum, void *table,
                             uint32_t capacity, void *stream)
{
    return hash_build_impl(coords, n, nullptr, quantum, table, capacity, stream);
}

int eprecon_hash_build_dn_async(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, int quantum, void *table,
                                uint32_t capacity, void *stream)
{
    if (!n_dev) return EPRECON_ERR_ARG;
    return hash_build_impl(coords, n_cap, n_dev, quantum, table, capacity, stream);
}

int eprecon_hash_query_async(const void *table, uint32_t capacity, const int32_t *queries, int64_t m,
                             int quantum, int32_t *out_index, void *stream)
{
    if (!table || !is_pow2(capacity) || m < 0 || quantum < 1 || (m > 0 && (!queries || !out_index)))
        return EPRECON_ERR_ARG;
    if (m == 0) return EPRECON_OK;
    HashTable t = make_table(const_cast<void *>(table), capacity);
    hipLaunchKernelGGL(hash_query_kernel, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0,
                       (hipStream_t)stream, t, reinterpret_cast<const int4 *>(queries), (int)m, quantum,
                       out_index);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_hash_status(const void *table, void *stream)
{
    int32_t s = 0;
    EP_HIP_CHECK(hipMemcpyAsync(&s, table, sizeof(s), hipMemcpyDeviceToHost, (hipStream_t)stream));
    EP_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return s == 0 ? EPRECON_OK : (s & 1 ? EPRECON_ERR_UNSUPPORTED : EPRECON_ERR_WORKSPACE);
}

/* out[i] = sum of in[0 .. i) (int32, n < 2^31); *total (optional) = the grand total; scratch: ceil(n / 2048) int32 (used by the
 * two-launch form only).  The device-wide scan every compaction / numbering of the path runs on: one workgroup up to 32,768
 * elements, one decoupled look-back launch up to 8.4 M. */
int eprecon_exclusive_scan_async(const int32_t *in, int64_t n, int32_t *out, int32_t *total, int32_t *scratch, void *stream)
{
    if (n < 0 || n > 0x7fffffff || (n > 0 && (!in || !out || !scratch))) return EPRECON_ERR_ARG;
    return ep::exclusive_scan_i32(in, (int)n, out, scratch, total, (hipStream_t)stream);
}

size_t eprecon_unique_workspace_bytes(int64_t n)
{
    return align_up((size_t)(n > 0 ? n : 1) * 4, 256) * 3 + align_up((size_t)ceil_div(n > 0 ? n : 1, 2048) * 4, 256) + 256;
}

static int unique_coords_impl(const int32_t *coords, int64_t n, const int32_t *n_dev, int quantum, void *table,
                              uint32_t capacity, int32_t *inverse, int32_t *unique_coords, int32_t *n_unique_dev,
                              void *workspace, size_t workspace_bytes, void *stream, bool cleared = false,
                              int32_t *status_copy = nullptr)
{
    if (!n_unique_dev || !workspace || workspace_bytes < eprecon_unique_workspace_bytes(n))
        return n_unique_dev && workspace ? EPRECON_ERR_WORKSPACE : EPRECON_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc = hash_build_impl(coords, n, n_dev, quantum, table, capacity, stream, cleared);
    if (rc != EPRECON_OK) return rc;
    if (n == 0) {
        EP_HIP_CHECK(hipMemsetAsync(n_unique_dev, 0, sizeof(int32_t), st));
        if (status_copy) EP_HIP_CHECK(hipMemsetAsync(status_copy, 0, sizeof(int32_t), st));
        return EPRECON_OK;
    }
    if (!inverse || !unique_coords) return EPRECON_ERR_ARG;
    char *ws = reinterpret_cast<char *>(workspace);
    const size_t seg = align_up((size_t)n * 4, 256);
    int32_t *first = reinterpret_cast<int32_t *>(ws);
    int32_t *owner = reinterpret_cast<int32_t *>(ws + seg);
    int32_t *rank = reinterpret_cast<int32_t *>(ws + 2 * seg);
    int32_t *scratch = reinterpret_cast<int32_t *>(ws + 3 * seg);
    HashTable t = make_table(table, capacity);
    const dim3 grid((unsigned)ceil_div(n, 256)), block(256);
    const int4 *c4 = reinterpret_cast<const int4 *>(coords);
    hipLaunchKernelGGL(first_flag_kernel, grid, block, 0, st, t, c4, (int)n, quantum, first, owner, n_dev);
    EP_LAUNCH_CHECK();
    rc = ep::exclusive_scan_i32_dn(first, (int)n, n_dev, rank, scratch, n_unique_dev, st);
    if (rc != EPRECON_OK) return rc;
    const int64_t span = n > (int64_t)capacity ? n : (int64_t)capacity;
    hipLaunchKernelGGL(unique_finalize_kernel, dim3((unsigned)ceil_div(span, 256)), block, 0, st, t, c4, (int)n, quantum, owner, rank,
                       inverse, reinterpret_cast<int4 *>(unique_coords), n_dev, capacity, (const int32_t *)table_status(table),
                       status_copy);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"

namespace ep {
int unique_coords_dn(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, int quantum, void *table, uint32_t capacity,
                     int32_t *inverse, int32_t *unique_coords, int32_t *n_unique_dev, void *workspace, size_t workspace_bytes,
                     bool table_cleared, int32_t *status_copy, void *stream)
{
    if (!n_dev) return EPRECON_ERR_ARG;
    return unique_coords_impl(coords, n_cap, n_dev, quantum, table, capacity, inverse, unique_coords, n_unique_dev, workspace,
                              workspace_bytes, stream, table_cleared, status_copy);
}
}  // namespace ep

extern "C" {

int eprecon_unique_coords_async(const int32_t *coords, int64_t n, int quantum, void *table,
                                uint32_t capacity, int32_t *inverse, int32_t *unique_coords,
                                int32_t *n_unique_dev, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    return unique_coords_impl(coords, n, nullptr, quantum, table, capacity, inverse, unique_coords, n_unique_dev, workspace,
                              workspace_bytes, stream);
}

int eprecon_unique_coords_dn_async(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, int quantum, void *table,
                                   uint32_t capacity, int32_t *inverse, int32_t *unique_coords, int32_t *n_unique_dev,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    if (!n_dev) return EPRECON_ERR_ARG;
    return unique_coords_impl(coords, n_cap, n_dev, quantum, table, capacity, inverse, unique_coords, n_unique_dev, workspace,
                              workspace_bytes, stream);
}

__global__ void gather_headers_kernel(const int32_t *t0, const int32_t *t1, const int32_t *t2, int levels, int32_t *out)
{
    const int i = threadIdx.x;      // (status, count) of every level's table, side by side: ONE device -> host read later
    if (i < 2 * levels) out[i] = (i < 2 ? t0 : i < 4 ? t1 : t2)[i & 1];
}

int eprecon_unique_hierarchy_dn_async(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, int levels, void *const *tables,
                                      const uint32_t *capacities, int32_t *const *inverse, int32_t *const *unique_coords,
                                      int32_t *summary, void *workspace, size_t workspace_bytes, void *stream)
{
    if (levels < 1 || levels > 3 || !tables || !capacities || !inverse || !unique_coords || n_cap < 0) return EPRECON_ERR_ARG;
    ep::FillRegion reg[9];
    for (int l = 0; l < levels; ++l) {
        const int rc = ep::table_clear_regions(tables[l], capacities[l], reg + 3 * l);
        if (rc != EPRECON_OK) return rc;
    }
    int rc = ep::multi_fill(reg, 3 * levels, (hipStream_t)stream);     // ONE launch resets the tables of all strides
    if (rc != EPRECON_OK) return rc;
    const int32_t *src = coords, *live = n_dev;
    for (int l = 0; l < levels; ++l) {
        int32_t *count = reinterpret_cast<int32_t *>(tables[l]) + 1;       // next to the table's status word: one 8-byte read per stride
        rc = unique_coords_impl(src, n_cap, live, 1 << l, tables[l], capacities[l], inverse[l], unique_coords[l], count, workspace,
                                workspace_bytes, stream, true, nullptr);
        if (rc != EPRECON_OK) return rc;
        src = unique_coords[l];
        live = count;
    }
    if (summary) {
        hipLaunchKernelGGL(gather_headers_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int32_t *)tables[0],
                           (const int32_t *)tables[levels > 1 ? 1 : 0], (const int32_t *)tables[levels > 2 ? 2 : 0], levels, summary);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}

int eprecon_kernel_map_async(const void *table, uint32_t capacity, const int32_t *coords, int64_t n,
                             int ksize, int stride, int32_t *nbr, void *stream)
{
    if (!table || !is_pow2(capacity) || n < 0 || (ksize != 2 && ksize != 3) || stride < 1 ||
        (n > 0 && (!coords || !nbr)))
        return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    HashTable t = make_table(const_cast<void *>(table), capacity);
    const int kvol = ksize * ksize * ksize;
    hipLaunchKernelGGL(kernel_map_kernel, dim3((unsigned)ceil_div(n, 256), kvol), dim3(256), 0,
                       (hipStream_t)stream, t, reinterpret_cast<const int4 *>(coords), (int)n, ksize,
                       stride, nbr);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"

namespace ep {
// the region of a self map that must hold -1 before kernel_map_self runs: offsets 14 .. 26
FillRegion kernel_map_self_fill_region(int32_t *nbr, int64_t n)
{
    return FillRegion{nbr + (size_t)14 * n, (size_t)13 * n * sizeof(int32_t), 0xFFFFFFFFu};
}
}  // namespace ep

namespace {
int kernel_map_self_impl(const void *table, uint32_t capacity, const int32_t *coords, int64_t n, int stride, int32_t *nbr,
                         bool prefilled, void *stream)
{
    if (!table || !is_pow2(capacity) || n < 0 || stride < 1 || (n > 0 && (!coords || !nbr))) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    HashTable t = make_table(const_cast<void *>(table), capacity);
    if (!prefilled)
        EP_HIP_CHECK(hipMemsetAsync(nbr + (size_t)14 * n, 0xFF, (size_t)13 * n * sizeof(int32_t), (hipStream_t)stream));
    hipLaunchKernelGGL(kernel_map_self_kernel, dim3((unsigned)ceil_div(n, 256), 14), dim3(256), 0, (hipStream_t)stream, t,
                       reinterpret_cast<const int4 *>(coords), (int)n, stride, nbr);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}
}  // namespace

namespace ep {
int kernel_map_self_prefilled(const void *table, uint32_t capacity, const int32_t *coords, int64_t n, int stride, int32_t *nbr,
                              void *stream)
{
    return kernel_map_self_impl(table, capacity, coords, n, stride, nbr, true, stream);
}
}  // namespace ep

extern "C" {

int eprecon_kernel_map_self_async(const void *table, uint32_t capacity, const int32_t *coords, int64_t n, int stride, int32_t *nbr,
                                  void *stream)
{
    return kernel_map_self_impl(table, capacity, coords, n, stride, nbr, false, stream);
}


int eprecon_transpose_map_async(const int32_t *fine_coords, int64_t n, const int32_t *parent,
                                int fine_stride, int32_t *up_map, void *stream)
{
    if (n < 0 || fine_stride < 1 || (n > 0 && (!fine_coords || !parent || !up_map))) return EPRECON_ERR_ARG;
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(transpose_map_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const int4 *>(fine_coords), (int)n, parent,
                       fine_stride, up_map);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_grid_rank_async(const int32_t *coords, int64_t n, int stride, int grid_x, int grid_y, int grid_z, int32_t *rank,
                            void *stream)
{
    if (n < 0 || stride < 1 || grid_x <= 0 || grid_y <= 0 || grid_z <= 0 || !rank || (n > 0 && !coords) ||
        (int64_t)grid_x * grid_y * grid_z > 0x7ffffff0 || n > 0x7fffffff)
        return EPRECON_ERR_ARG;
    // rank[cells] then one int32: voxels off the grid (must stay 0; the caller may read it back)
    const size_t cells = (size_t)grid_x * grid_y * grid_z;
    EP_HIP_CHECK(hipMemsetAsync(rank, 0xff, cells * sizeof(int32_t), (hipStream_t)stream));
    EP_HIP_CHECK(hipMemsetAsync(rank + cells, 0, sizeof(int32_t), (hipStream_t)stream));
    if (n == 0) return EPRECON_OK;
    hipLaunchKernelGGL(grid_rank_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const int4 *>(coords), (int)n, stride, grid_x, grid_y, grid_z, rank, rank + cells);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

int eprecon_pixel_map_async(int maps, int height, int width, int ksize, int32_t *nbr, void *stream)
{
    if (maps <= 0 || height <= 0 || width <= 0 || ksize < 1 || ksize > 7 || !(ksize & 1) || !nbr ||
        (int64_t)maps * height * width > 0x7fffffff)
        return EPRECON_ERR_ARG;
    const int n = maps * height * width;
    hipLaunchKernelGGL(pixel_map_kernel, dim3((unsigned)ceil_div(n, 256), ksize * ksize), dim3(256), 0,
                       (hipStream_t)stream, maps, height, width, ksize, nbr);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // extern "C"
