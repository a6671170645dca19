// Open-addressing hash grid over packed voxel keys (device side), shared by the kernel-map,
// voxelise / devoxelise and union kernels.
//
// Replaces torchsparse's F.sphash / F.sphashquery (call sites ops/torchsparse_utils.py:19-21,44-50,
// 73-79 of the reference) and the coordinate bookkeeping spconv does internally.  The hash value
// never leaves the library, so any collision-free key works: a key is the
// exact 64-bit packing of (batch, x, y, z), the table stores the key itself, and a lookup compares
// keys — there are no false positives.
//
//   key  = batch[4 bits] | (x + 2^19)[20] | (y + 2^19)[20] | (z + 2^19)[20]
//   slot = mix64(key) & (capacity - 1), linear probing, capacity = power of two >= 2 * n
//   value = smallest row index that inserted the key (atomicMin) -> duplicates collapse
//           deterministically onto their first occurrence.
#pragma once

#include "common.hpp"

namespace ep {

constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr int kCoordBias = 1 << 19;
constexpr int kCoordLimit = (1 << 19) - 1;  // |coordinate| must stay below this
constexpr int kMaxBatch = 14;               // batch 15 is reserved (all-ones key = empty slot)

struct HashTable {
    unsigned long long *keys;  // [capacity]
    int32_t *vals;             // [capacity]
    uint32_t mask;             // capacity - 1
};

__host__ __device__ __forceinline__ bool key_in_range(int b, int x, int y, int z)
{
    return b >= 0 && b <= kMaxBatch && x > -kCoordLimit && x < kCoordLimit && y > -kCoordLimit &&
           y < kCoordLimit && z > -kCoordLimit && z < kCoordLimit;
}

__host__ __device__ __forceinline__ unsigned long long pack_key(int b, int x, int y, int z)
{
    return ((unsigned long long)(unsigned)b << 60) |
           ((unsigned long long)(unsigned)(x + kCoordBias) << 40) |
           ((unsigned long long)(unsigned)(y + kCoordBias) << 20) |
           (unsigned long long)(unsigned)(z + kCoordBias);
}

__device__ __forceinline__ uint32_t hash_slot(unsigned long long k, uint32_t mask)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k & mask;
}

// insert (key -> min(row)); returns false if the probe sequence wrapped (table full)
__device__ __forceinline__ bool hash_insert(const HashTable &t, unsigned long long key, int row)
{
    uint32_t s = hash_slot(key, t.mask);
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        const unsigned long long prev = atomicCAS(&t.keys[s], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) {
            atomicMin(&t.vals[s], row);
            return true;
        }
        s = (s + 1) & t.mask;
    }
    return false;
}

// lookup after the build kernel has completed (kernel boundary = visibility); -1 when absent
__device__ __forceinline__ int hash_lookup(const HashTable &t, unsigned long long key)
{
    uint32_t s = hash_slot(key, t.mask);
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        const unsigned long long k = t.keys[s];
        if (k == key) return t.vals[s];
        if (k == kEmptyKey) return -1;
        s = (s + 1) & t.mask;
    }
    return -1;
}

static inline uint32_t hash_capacity_for(int64_t n)
{
    uint64_t c = 1024;
    while (c < (uint64_t)(2 * n + 1)) c <<= 1;
    return (uint32_t)c;
}

}  // namespace ep
