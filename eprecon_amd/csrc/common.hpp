// Shared device/host helpers for libeprecon_hip.so (gfx950 only; wave = 64, 8 XCDs).
//
// The whole library is compiled with -ffp-contract=off: wherever a fused multiply-add is part of
// the arithmetic contract (or wanted for speed) it is written as fmaf()/__fmaf_rn explicitly, and
// nowhere else may the compiler fuse — bit-exact voxel indices depend on it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/eprecon_hip.h"

#define EP_HIP_CHECK(expr)                                            \
    do {                                                              \
        hipError_t e_ = (expr);                                       \
        if (e_ != hipSuccess) return EPRECON_ERR_HIP_BASE - (int)e_;  \
    } while (0)
#define EP_LAUNCH_CHECK() EP_HIP_CHECK(hipGetLastError())

namespace ep {

constexpr int kWave = 64;
constexpr int kXcd = 8;

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Hardware block id -> logical block id.  The dispatcher places hardware block h on XCD h % 8
// (observed, speed only); this bijection gives every XCD one contiguous range of logical blocks
// so that neighbouring tiles, which touch neighbouring image regions, share one L2.
__device__ __forceinline__ int xcd_remap(int hw_bid, int nblk)
{
    const int q = nblk / kXcd, r = nblk % kXcd;
    const int xcd = hw_bid % kXcd, k = hw_bid / kXcd;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// Exclusive rank of this thread among the threads of its block for which `pred` holds, by wave
// ballot + popcount and a per-wave count table in LDS (`wave_counts` holds BLOCK/64 ints).
// Contains one __syncthreads(); every thread of the block must call it.
template <int BLOCK>
__device__ __forceinline__ int block_exclusive_rank(bool pred, int *wave_counts, int &block_total)
{
    const unsigned long long m = __ballot(pred);
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x / kWave;
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_counts[wid] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / kWave; ++w) {
        const int c = wave_counts[w];
        off += (w < wid) ? c : 0;
        tot += c;
    }
    block_total = tot;
    return off + within;
}

// device-wide exclusive scan of int32 (kernel_map.hip); scratch holds ceil(n / 2048) int32
int exclusive_scan_i32(const int32_t *in, int n, int32_t *out, int32_t *scratch, int32_t *total_dev,
                       hipStream_t st);
// the same with the live length on the device: min(n, *n_dev) (n_dev may be nullptr); launches are sized by n
int exclusive_scan_i32_dn(const int32_t *in, int n, const int32_t *n_dev, int32_t *out, int32_t *scratch, int32_t *total_dev,
                          hipStream_t st);

// Several device regions filled with a 32-bit pattern each in ONE launch (instead of a hipMemsetAsync / clear kernel per
// region): the prologues of the stage calls clear a handful of index volumes, counters and hash tables of a few MB each, and
// every launch they do not need is ~5 us of a fragment's chain.  Regions: 4-byte aligned, sizes multiples of 4 bytes.
struct FillRegion {
    void *p;
    size_t bytes;
    uint32_t value;
};
constexpr int kMaxFillRegions = 12;
int multi_fill(const FillRegion *regions, int count, hipStream_t st);
// the three regions that reset an open-addressing table (header, keys = empty, values = int max): kernel_map.hip
int table_clear_regions(void *table, uint32_t capacity, FillRegion *out3);

// eprecon_kernel_map_self_async whose caller filled the map's upper half (offsets 14 .. 26) with -1 itself — the region
// kernel_map_self_fill_region names, typically inside a multi_fill that resets other things too
FillRegion kernel_map_self_fill_region(int32_t *nbr, int64_t n);
int kernel_map_self_prefilled(const void *table, uint32_t capacity, const int32_t *coords, int64_t n, int stride, int32_t *nbr,
                              void *stream);

// eprecon_trilinear_map_async without hash probes (csrc/voxelize.hip): base_row[i] = row of point i's base voxel in the set whose
// 3x3x3 kernel map nbr27 int32[27][m] is given; the corners are that voxel's neighbours at the offsets {0, +1}^3
int trilinear_from_map(const float *points_xyzb, int64_t n, const int32_t *base_row, const int32_t *nbr27, int64_t m, int stride,
                       int32_t *idx8, float *weight8, void *stream);

// eprecon_unique_coords_dn_async for callers inside the library: table_cleared = the caller reset the table itself (multi_fill);
// status_copy (optional device int32): receives the table's status word from the call's last launch
int unique_coords_dn(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, int quantum, void *table, uint32_t capacity,
                     int32_t *inverse, int32_t *unique_coords, int32_t *n_unique_dev, void *workspace, size_t workspace_bytes,
                     bool table_cleared, int32_t *status_copy, void *stream);

// A side stream of the library's own per caller stream, for entry points that issue two independent chains of small launches
// (csrc/gru_stage_finish.hip): record ev_fork on the caller's stream and let `side` wait for it, issue the second chain on
// `side`, record ev_join there and let the caller's stream wait for it.
struct Fork {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};
int fork_for(hipStream_t main, Fork &out);

// the gather-kernel timing hook of eprecon_profile_enable (back_project.hip), for the other gather variants
int profile_bracket_begin(hipStream_t st);
int profile_bracket_end(hipStream_t st, const char *kernel);

}  // namespace ep
