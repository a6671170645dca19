// Direct gather form of the 3x3x3 sparse convolution on v_mfma_f32_16x16x4_f32 for long lists (C_out <= 64): NO operand goes
// through LDS and there is no barrier in the loop, so the waves of a CU drift apart instead of staging, gathering and
// multiplying in lockstep (the phases of the LDS-resident kernel add up).
//   A operand (lane l: row l & 15, k index q = l >> 4): one 16-byte buffer gather x[nbr[k][row]][16 kc + 4 q .. + 3] per
//     (offset, 16-channel chunk) feeds four MFMAs; a missing neighbour is sent past the end of the buffer (zeros).
//   B operand: the weights pre-packed by pack_weights16_kernel in operand order (wq16), one coalesced 1 KB buffer load per
//     (offset, chunk, 16-column tile), served by L1 / L2 — every wave of the launch walks the same sequence.
//   A wave owns 2 x 16 rows and all CT column tiles (2 x CT accumulators of 4 VGPRs); the loads of the next stage (1..3
//   chunks) are in flight while the current one runs its MFMAs, unconditionally, so the waits are `vmcnt(<loads of a stage>)`.
// Timing ablations (profiles/r03/conv_direct_ablate.txt): with the gathers switched off and every weight load an L1 hit the
// first version of this kernel lost 8 % of its time — not memory but the instruction stream around the MFMAs (run-time
// (offset, chunk) arithmetic, masks) kept the matrix pipe at ~70 %: a 16x16x4 MFMA is 32 cycles, eight issue slots.  The
// chunk count is therefore a template parameter: offsets inside a stage are instruction immediates, one multiply + select
// per (offset, row tile), nothing else between the MFMAs.
// Summation order differs from the 32x32x2 kernels: equal within fp32 round-off, not bit for bit.
// (Included by sparse_conv_direct.hip — eligibility rule and dispatch — and by sparse_conv_direct_ct{1..4}.hip, which instantiate the
// kernels of one column-tile count each: four translation units compile side by side instead of one for two minutes.)
#pragma once
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.hpp"
#include "conv_common.hpp"

namespace epconv {
namespace {
using namespace ep;

constexpr int kRT = 2;   // 16-row tiles per wave

// Timing ablations of the probe builds only (python -m eprecon_amd.build --variant X -DEP_DIRECT_ABL=N sparse_conv_direct.hip;
// 0 in the library): 2 every gather reads one of 16 L1-resident rows, 4 no gathers, 8 every weight load reads the first
// (offset, chunk) block, 16 no MFMAs (the operands are summed on the VALU instead, so the loads stay), 32 no weight loads.
// Wrong results by design.
#ifndef EP_DIRECT_ABL
#define EP_DIRECT_ABL 0
#endif

// ReLU of a gathered value without the canonicalising `v_max_f32 x, x, x` clang puts in front of every llvm.maxnum (it must
// quiet signalling NaNs): 8 extra VALU instructions per gathered quad next to its 8 MFMAs.  The instruction itself is written
// out: v_max_f32 returns the non-NaN operand, so ReLU(NaN) = 0 as with fmaxf.  Rounds 4-5 got the same instruction count from
// a translation-unit-wide -fno-honor-nans (and the compiler folds v_med3_f32(x, 0, inf) back into maxnum + canonicalise);
// this keeps every other floating-point operation of the file under the strict rules.
__device__ __forceinline__ float relu_nc(float x)
{
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
// ... chosen by a launch-uniform bit mask instead of a branch (all ones: ReLU, zero: x unchanged — bit for bit what the branch gives)
__device__ __forceinline__ float relu_sel(float x, unsigned mask)
{
    const unsigned r = __float_as_uint(relu_nc(x)), b = __float_as_uint(x);
    return __uint_as_float((r & mask) | (b & ~mask));
}

// ---- opt-in bf16x3 operands (EPRECON_CONV_BF16X3=1; DESIGN.md 3b, SURVEY.md section 7 "with an error budget") ---------------
// x = hi + lo + e with hi = the top 16 bits of x (a bf16 by truncation), lo = bf16(x - hi) (round to nearest even; x - hi is exact)
// and |e| <= 2^-17 |x|.  A product a * w is taken as a_hi w_hi + a_hi w_lo + a_lo w_hi on the bf16 matrix pipe (fp32 accumulate):
// what is dropped is a_lo w_lo and the two e terms, <= ~2^-15 |a w| in all against fp32's 2^-24.  Both operands are split in
// registers from the SAME fp32 gathers and the same operand-order packing (wq16) the fp32 path loads: no second weight format.
// The 16-byte quad of a lane (four consecutive input channels, the k slot of v_mfma_f32_16x16x4_f32) is also the lane's four k
// values of v_mfma_f32_16x16x16_bf16, and two quads side by side are its eight of v_mfma_f32_16x16x32_bf16 — A and B agree on
// which channel sits in which k position, which is all a dot product over k needs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct BfQuad { unsigned hi[2], lo[2]; };       // four values as two packed bf16 pairs each
__device__ __forceinline__ BfQuad bf_split(const float4 &v)
{
    const unsigned b0 = __float_as_uint(v.x), b1 = __float_as_uint(v.y), b2 = __float_as_uint(v.z), b3 = __float_as_uint(v.w);
    BfQuad r;
    r.hi[0] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);       // (upper halves of two registers side by side)
    r.hi[1] = __builtin_amdgcn_perm(b3, b2, 0x07060302u);
    const f32x2 r01 = {v.x - __uint_as_float(b0 & 0xffff0000u), v.y - __uint_as_float(b1 & 0xffff0000u)};
    const f32x2 r23 = {v.z - __uint_as_float(b2 & 0xffff0000u), v.w - __uint_as_float(b3 & 0xffff0000u)};
    r.lo[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));      // v_cvt_pk_bf16_f32
    r.lo[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
    return r;
}
__device__ __forceinline__ f32x4 bf_mfma32(const unsigned (&a0)[2], const unsigned (&a1)[2], const unsigned (&b0)[2],
                                           const unsigned (&b1)[2], f32x4 acc)
{
    union { bf16x8 v; unsigned u[4]; } a, b;
    a.u[0] = a0[0]; a.u[1] = a0[1]; a.u[2] = a1[0]; a.u[3] = a1[1];
    b.u[0] = b0[0]; b.u[1] = b0[1]; b.u[2] = b1[0]; b.u[3] = b1[1];
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 bf_mfma16(const unsigned (&a0)[2], const unsigned (&b0)[2], f32x4 acc)
{
    union { s16x4 v; unsigned u[2]; } a, b;
    a.u[0] = a0[0]; a.u[1] = a0[1];
    b.u[0] = b0[0]; b.u[1] = b0[1];
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.v, b.v, acc, 0, 0, 0);
}

// Epilogue for the 16x16 accumulator layout (column l & 15, rows 4 (l >> 4) + reg): bias, ReLU, residual with its pending
// BatchNorm, row-wise LayerNorm (16-lane xor-shuffles), BatchNorm summaries of the 128-row block (fixed-order Chan merges: rows
// in the lane, lane groups, waves).
// PERWAVE (the persistent kernel below): row0 is the first of the WAVE's 32 rows and `block` its job index; the wave's own
// summary (count, mean, M2 of 32 rows) is the partial row — no workgroup merge, no barrier.
template <int CT, bool PERWAVE = false>
__device__ __forceinline__ void direct_epilogue(const ConvParams &p, f32x4 (&acc)[kRT][CT], int row0, float *sStat, int block = 0)
{
    constexpr int RT = kRT, NR = 4 * kRT;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = PERWAVE ? 0 : tid >> 6;
    const int l16 = lane & 15, q = lane >> 4;
    int orow[NR];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = row0 + wave * 16 * RT + 16 * rt + 4 * q + j;
            orow[4 * rt + j] = row < p.n_out ? row : -1;
        }
    float v[CT][NR];
    bool colok[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int col = 16 * t + l16;
        colok[t] = col < p.Cout;
        const float b = (p.bias && colok[t]) ? p.bias[col] : 0.0f;
        const float rs = (p.res_scale && colok[t]) ? p.res_scale[col] : 1.0f;
        const float rb = (p.res_scale && colok[t]) ? p.res_shift[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float val = 0.0f;
            if (colok[t] && orow[r] >= 0) {
                val = acc[r >> 2][t][r & 3] + b;
                if (p.relu) val = fmaxf(val, 0.0f);
                if (p.res) {
                    float rv = p.res[(size_t)orow[r] * p.ld_res + col];
                    if (p.res_scale) {
                        rv = fmaf(rv, rs, rb);
                        if (p.res_relu) rv = fmaxf(rv, 0.0f);
                    }
                    val += rv;
                }
            }
            v[t][r] = val;
        }
    }
    if (p.ln) {  // (uniform) row-wise LayerNorm over the C_out columns: 16 lanes x CT tiles hold a row
        const float inv_c = 1.0f / (float)p.Cout;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float sum = 0.0f;
#pragma unroll
            for (int t = 0; t < CT; ++t) sum += v[t][r];
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
            const float mean = sum * inv_c;
            float sq = 0.0f;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const float d = colok[t] ? v[t][r] - mean : 0.0f;
                v[t][r] = d;
                sq = fmaf(d, d, sq);
            }
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) sq += __shfl_xor(sq, m);
            const float inv = 1.0f / sqrtf(sq * inv_c + p.ln_eps);
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int col = 16 * t + l16;
                float y = fmaf(v[t][r] * inv, (p.ln_gamma && colok[t]) ? p.ln_gamma[col] : 1.0f, (p.ln_beta && colok[t]) ? p.ln_beta[col] : 0.0f);
                if (p.ln_post_relu) y = fmaxf(y, 0.0f);
                v[t][r] = y;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (colok[t] && orow[r] >= 0) p.out[(size_t)orow[r] * p.ld_out + 16 * t + l16] = v[t][r];
    if (p.bn_partial || p.bn_acc) {  // (uniform) (count, mean, M2) of the stored values per column: rows in the lane, lane groups, waves
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            float n = 0.0f, sum = 0.0f;
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (orow[r] >= 0) { n += 1.0f; sum += v[t][r]; }
            float mean = n > 0.0f ? sum / n : 0.0f, m2 = 0.0f;
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (orow[r] >= 0) { const float d = v[t][r] - mean; m2 = fmaf(d, d, m2); }
#pragma unroll
            for (int m = 16; m < 64; m <<= 1) {  // lane groups q in order: the lower group is the left operand
                const float on = __shfl_xor(n, m), om = __shfl_xor(mean, m), oq = __shfl_xor(m2, m);
                const bool lower = (lane & m) == 0;
                float a_n = lower ? n : on, a_mean = lower ? mean : om, a_m2 = lower ? m2 : oq;
                chan_merge(a_n, a_mean, a_m2, lower ? on : n, lower ? om : mean, lower ? oq : m2);
                n = a_n; mean = a_mean; m2 = a_m2;
            }
            if constexpr (PERWAVE) {
                if (q == 0 && colok[t]) {
                    if (p.bn_partial) {
                        float *dst = p.bn_partial + (size_t)block * 3 * p.Cout + 16 * t + l16;
                        dst[0] = n; dst[p.Cout] = mean; dst[2 * p.Cout] = m2;
                    }
                    if (p.bn_acc) bn_acc_publish(p, 16 * t + l16, block, block == 0, n, mean, m2);
                }
                continue;
            }
            if (q == 0) {
                float *d = sStat + (wave * 3) * 16 * CT + 16 * t + l16;
                d[0] = n; d[16 * CT] = mean; d[2 * 16 * CT] = m2;
            }
        }
        if constexpr (PERWAVE) return;
        __syncthreads();
        if (tid < 16 * CT && tid < p.Cout) {
            float a_n = 0.0f, a_mean = 0.0f, a_m2 = 0.0f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w)
                chan_merge(a_n, a_mean, a_m2, sStat[(w * 3) * 16 * CT + tid], sStat[(w * 3 + 1) * 16 * CT + tid], sStat[(w * 3 + 2) * 16 * CT + tid]);
            if (p.bn_partial) {
                float *dst = p.bn_partial + (size_t)blockIdx.x * 3 * p.Cout + tid;
                dst[0] = a_n; dst[p.Cout] = a_mean; dst[2 * p.Cout] = a_m2;
            }
            if (p.bn_acc) bn_acc_publish(p, tid, (int)blockIdx.x, blockIdx.x == 0, a_n, a_mean, a_m2);
        }
    }
}

// C_out = 16 m + 8 (8, 24, 40, 56: the ConvGRU / SPVCNN channel plans at the finest level): the last 8 columns run on
// v_mfma_f32_4x4x1_16B_f32 instead of a half-empty 16-column tile.  That instruction multiplies 16 independent 4x1 by 1x4
// blocks (lane 4 b + i holds A_b[i] / B_b[i], register r of lane 4 b + j holds D_b[r][j]); fed with the SAME A register as the
// 16x16x4 MFMA (lane = row l & 15, k slot g = l >> 4), block b = 4 g + (row >> 2) is rows 4 (row >> 2) .. + 3 at the ONE input
// channel of k slot g, so with B_b[j] = W[channel of slot g][column 16 m + 4 cg + j] one instruction adds a k slot's product
// into a per-slot partial of 16 rows x 4 columns: 512 flops in 8 cycles, the fp32 rate of every MFMA shape.  Two of them (cg =
// 0, 1: 16 cycles) replace one 16x16x4 (32 cycles) per gathered register; the four k-slot partials are summed once, here, in
// fixed order, and handed to the shared epilogue in the 16x16 accumulator layout (column l & 15, rows 4 (l >> 4) + reg;
// columns 8 .. 15 zero: the epilogue masks them by C_out anyway).
__device__ __forceinline__ void tail_to_tile(const f32x4 (&acct)[kRT][2], f32x4 (&dst)[kRT], int lane)
{
    const int l16 = lane & 15, q = lane >> 4;
    const int src = 4 * q + (l16 & 3);      // k slot 0's lane of (rows 4 q .. 4 q + 3, column l16 & 3): after the sums every slot holds the total
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v0 = acct[rt][0][j], v1 = acct[rt][1][j];
            v0 += __shfl_xor(v0, 16); v1 += __shfl_xor(v1, 16);
            v0 += __shfl_xor(v0, 32); v1 += __shfl_xor(v1, 32);
            const float t0 = __shfl(v0, src), t1 = __shfl(v1, src);
            dst[rt][j] = l16 < 4 ? t0 : (l16 < 8 ? t1 : 0.0f);
        }
}

// The workgroup's slice of the kernel map -> LDS ([K][128 rows]), and for every wave the offsets at which at least one of its
// 32 rows has a neighbour, as a bit mask (wave-uniform: it lives in scalar registers, and the walk below — lowest set bit,
// clear it — is scalar arithmetic).  All loads of a thread are issued before the first is used (K <= 27: at most 14 per
// thread).  Load pass `it` of wave w covers offset 2 it + w / 2, rows 64 (w % 2) .. + 63: the two halves of its ballot are
// the flags of consumer waves 2 (w % 2) and 2 (w % 2) + 1, left in sFlag[offset][consumer wave].  Ends with the barrier that
// publishes the table (and whatever the caller staged before the call).
constexpr int kMapLoads = 14;
__device__ __forceinline__ unsigned stage_map(const ConvParams &p, int row0, int *sNbr, int *sFlag, int tid)
{
    static_assert(kDirectRows == 128 && kRT == 2 && kWaves == 4, "the pass -> (offset, row half) arithmetic below");
    const int total = p.K * kDirectRows;
    const int lane = tid & 63, wave = tid >> 6;
    int jv[kMapLoads];
#pragma unroll
    for (int it = 0; it < kMapLoads; ++it) {
        const int e = tid + it * 256;
        const int k = e >> 7, row = row0 + (e & 127);
        int j = -1;
        if (e < total && row < p.n_out) j = p.nbr ? p.nbr[(size_t)k * p.n_out + row] : row;
        jv[it] = j;
    }
#pragma unroll
    for (int it = 0; it < kMapLoads; ++it) {
        const int e = tid + it * 256;
        if (e < total) sNbr[e] = jv[it];
        const unsigned long long b = __ballot(jv[it] >= 0);
        const int k = 2 * it + (wave >> 1);
        if (lane < 2 && k < p.K) sFlag[k * kWaves + 2 * (wave & 1) + lane] = (lane ? (unsigned)(b >> 32) : (unsigned)b) != 0u;
    }
    __syncthreads();
    const bool mine = lane < p.K && sFlag[lane * kWaves + wave] != 0;
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(mine));
}

// Position in the (live offset, part) sequence of a wave: `k` the current offset, `rest` the live offsets after it.
// next() stays on the last position once the sequence is exhausted (a stage fetched past the end is not consumed).
struct LiveCursor {
    unsigned rest;
    int k, part;
    __device__ __forceinline__ explicit LiveCursor(unsigned mask) : rest(mask & (mask - 1u)), k(__builtin_ctz(mask)), part(0) {}
    __device__ __forceinline__ void next(int parts)
    {
        if (part + 1 < parts) ++part;
        else if (rest) { k = __builtin_ctz(rest); rest &= rest - 1u; part = 0; }
    }
};

// EPRECON_CONV_INTERLEAVE=0: the fenced schedule for every launch (read per launch; see the main loop of the template kernel)
#ifndef EP_MIX_EXTRA       // (probe builds: matrix instructions per load group beyond ceil(MFMAs / loads) of a stage)
#define EP_MIX_EXTRA 1
#endif

// chunks per stage for a layer of KCH chunks: the stages of an offset are KCH / G
#ifndef EP_STAGE_CAP       // (probe builds: -DEP_STAGE_CAP=2 / 1 caps the chunks per stage — fewer registers, more waves per SIMD)
#define EP_STAGE_CAP 3
#endif
constexpr int stage_chunks(int kch) { return kch % 3 == 0 && EP_STAGE_CAP >= 3 ? 3 : (kch % 2 == 0 && EP_STAGE_CAP >= 2 ? 2 : 1); }

// TAIL: the last column tile holds <= 8 columns and runs on the 4x4x1 MFMAs (tail_to_tile above); its B operands are the tail
// section of the packing (pack_weights16_kernel: behind the CT tiles, 512 B per (offset, chunk)).
// BF: the bf16x3 operand form above (opt-in; padded 16-column tiles only: a half-empty tile costs 17 cycles there).
// MODE (the launcher's choice, launch_k): 0 any layer; 1 / 2 branch-free layers without / with a pending BatchNorm of the input;
// 3 / 4 the same for layers whose last chunk holds <= 8 channels (C_in = 8, 24, 40) when an offset is one stage
// (see `consume` and the main loop)
template <int CT, int KCH, bool TAIL = false, int G = stage_chunks(KCH), bool BF = false, int MODE = 0>
__global__ __launch_bounds__(256) void spconv_direct16_kernel(ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RT = kRT, ROWS = kDirectRows;
    constexpr int CTM = TAIL ? CT - 1 : CT;          // full 16-column tiles on the 16x16x4 MFMA
    constexpr int CTA = CTM > 0 ? CTM : 1;           // (array extent: no zero-length arrays)
    constexpr int PARTS = KCH / G;
    constexpr int cpad = 16 * KCH;
    int *sNbr = reinterpret_cast<int *>(smem);                        // [K][ROWS]
    float *sStat = reinterpret_cast<float *>(sNbr + p.K * ROWS);      // [4 waves][3][16 CT]
    float *sAff = sStat + kWaves * 3 * 16 * CT;                       // [2][cpad]
    int *sFlag = reinterpret_cast<int *>(sAff + 2 * cpad);            // [K][4 waves] the wave has a neighbour at the offset
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, q = lane >> 4;
    const int row0 = (int)blockIdx.x * ROWS;

    stage_in_affine<256>(p, sAff, cpad, tid);
    // Offsets none of the wave's 32 rows has a neighbour at are skipped altogether (no gathers, no weight loads, no MFMAs):
    // an output-stationary kernel otherwise multiplies zeros for every missing neighbour.  On the surface-shaped sets of a
    // fragment 12-25 % of the (32-row, offset) groups are dead; on the second voxelisation of ConvGRU's convr — already
    // scaled coordinates divided by the resolution again, models/modules.py:216-217: no two voxels are adjacent — 26 of the
    // 27 offsets are (profiles/r04/conv_tile_liveness.txt).
    const unsigned live = stage_map(p, row0, sNbr, sFlag, tid);

    // One full tile per row tile (C_out = 16, 24) leaves TWO independent 16x16x4 accumulators: consecutive MFMAs on one
    // accumulator issue 32 cycles apart but the result returns after 40 — a quarter of the matrix pipe's time in bubbles
    // (the compiler groups the 16x16x4s whatever order the source puts them in: ISA of round 6).  The channels of such a layer
    // are split over NS = 2 accumulator sets (components x, z / y, w of every gathered quad), summed once before the epilogue.
    static_assert(!(BF && TAIL), "the bf16x3 form runs on padded column tiles");
    static_assert(MODE < 3 || (G == KCH && !BF), "modes 3 / 4: one stage per offset, exact-fp32 form");

    constexpr int NS = (CTM == 1 && !BF) ? 2 : 1;
    f32x4 acc[NS][RT][CTA];
    f32x4 acct[RT][2];      // TAIL: per-k-slot partials of the last 8 columns (two groups of 4)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int t = 0; t < CTA; ++t) acc[ns][rt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        acct[rt][0] = acct[rt][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const unsigned row_bytes = (unsigned)p.ld_x * 4u, oob = (unsigned)p.x_bytes;
    const int U = __builtin_popcount(live) * PARTS;   // stages: (live offset, part)
    constexpr unsigned kChunkBytes = (unsigned)CT * 1024u;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wq16), 0, (int)((unsigned)p.K * KCH * kChunkBytes), 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    // tail section: [K][KCH][2 column groups][4 k slots][4 columns] float4 = 512 B per (offset, chunk)
    const __amdgpu_buffer_rsrc_t wtrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.wq16) + (size_t)p.K * KCH * CT * 256, 0, (int)((unsigned)p.K * KCH * 512u), 0x00020000);
    const unsigned tlane = (unsigned)(4 * q + (lane & 3)) * 16u;
    const int *myNbr = sNbr + wave * 16 * RT + l16;
    // a last chunk of <= 8 channels (C_in = 8, 24, 40): lane group q takes channels 2 q, 2 q + 1 and the chunk is two MFMAs
    const int last_c = p.Cin - 16 * (KCH - 1);      // channels of the last chunk
    const bool tail8 = last_c <= 8;
    const unsigned cq = 16u * (unsigned)q;
    const unsigned cq_last = tail8 ? 8u * (unsigned)q : cq;
    const bool last_ok = (tail8 ? 2 : 4) * q < last_c;

    struct Stage {
        float4 a[G][RT];
        float4 b[G][CTA];
        float4 bt[G][TAIL ? 2 : 1];
    };
    // stage u = (offset k = u / PARTS, chunks kc0 .. kc0 + G - 1 with kc0 = (u % PARTS) * G)
    auto fetch = [&](const LiveCursor &c, Stage &g) {
        const int k = c.k, part = c.part;                                // (wave-uniform: the weight offset below is a scalar)
        const unsigned xs = 64u * (unsigned)(part * G);                 // (scalar) byte offset of the stage's first chunk
        const unsigned ws = (EP_DIRECT_ABL & 8) ? 0u : (unsigned)(k * KCH + part * G) * kChunkBytes;
        const bool has_last = part == PARTS - 1;                         // (uniform) the stage holds the layer's last chunk
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            int j = myNbr[k * ROWS + 16 * rt];
            if (EP_DIRECT_ABL & 2) j = j >= 0 ? l16 : j;
            const unsigned rowsel = j >= 0 ? __umul24((unsigned)j, row_bytes) : oob;
            const unsigned v0 = rowsel + cq;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                unsigned off = v0 + 64u * i;
                if (i == G - 1 && has_last) off = last_ok ? rowsel + cq_last + 64u * i : oob;
                if (EP_DIRECT_ABL & 4) {
                    g.a[i][rt] = make_float4(__uint_as_float(off), 1.0f, 2.0f, 3.0f);
                    continue;
                }
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, xs, 0);
                g.a[i][rt] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
#pragma unroll
            for (int t = 0; t < CTM; ++t) {
                if (EP_DIRECT_ABL & 32) {
                    g.b[i][t] = make_float4(__uint_as_float(ws), 1.0f, 2.0f, 3.0f);
                    continue;
                }
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + (unsigned)t * 1024u, ws + (unsigned)i * kChunkBytes, 0);
                g.b[i][t] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
            if constexpr (TAIL && (EP_DIRECT_ABL & 32)) {
                g.bt[i][0] = g.bt[i][1] = make_float4(__uint_as_float(ws), 1.0f, 2.0f, 3.0f);
            } else if constexpr (TAIL) {
                const unsigned wts = (EP_DIRECT_ABL & 8) ? 0u : (unsigned)(k * KCH + part * G + i) * 512u;
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wtrsrc, tlane + (unsigned)cg * 256u, wts, 0);
                    g.bt[i][cg] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            }
        }
    };
    // MODE (compile time): 0 = every layer (launch-uniform branches on a pending BatchNorm of the input, an 8-channel or ragged
    // last chunk); 1 = C_in a multiple of 16... precisely: no 8-channel last chunk, no ragged count, no pending BatchNorm;
    // 2 = the same with a pending BatchNorm (+ ReLU, applied without a branch).  Modes 1 and 2 are ONE basic block: the main loop
    // below can then spread the next stage's loads among this stage's MFMAs.
    const unsigned relu_mask = p.in_relu ? 0xffffffffu : 0u;
    auto consume = [&](const LiveCursor &c, const Stage &g) {
        const int k = c.k, part = c.part;
        const bool has_last = part == PARTS - 1;
        BfQuad qa[BF ? G : 1][RT], qb[BF ? G : 1][CTA];                // BF: the stage's operands as bf16 (hi, lo) pairs
#pragma unroll
        for (int i = 0; i < G; ++i) {
            // (MODE 3 / 4: an 8-channel last chunk as a compile-time fact — the launcher takes them only when an offset is ONE stage,
            // PARTS == 1, so that the last chunk of every stage is the layer's last)
            const bool t8 = MODE >= 3 ? i == G - 1 : (MODE == 0 && tail8 && i == G - 1 && has_last);   // .z / .w of the gathered values are not used
            float4 av[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) av[rt] = g.a[i][rt];
            if (MODE == 0 && (p.Cin & 3) && i == G - 1 && has_last) {   // (uniform) ragged channel count on a padded pitch: the pad lanes stay out
                const int c = 16 * (KCH - 1) + (t8 ? 2 : 4) * q;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (c + 1 >= p.Cin) av[rt].y = 0.0f;
                    if (c + 2 >= p.Cin) av[rt].z = 0.0f;
                    if (c + 3 >= p.Cin) av[rt].w = 0.0f;
                }
            }
            if (MODE == 2 || MODE == 4 || (MODE == 0 && p.in_scale)) {  // (uniform) the producer's pending BatchNorm (+ ReLU) on the gathered values
                const int kc = part * G + i;
                const int ca = 16 * kc + (t8 ? 2 : 4) * q;
                const float4 sc = make_float4(sAff[ca], sAff[ca + 1], sAff[ca + 2], sAff[ca + 3]);
                const float4 sh = make_float4(sAff[cpad + ca], sAff[cpad + ca + 1], sAff[cpad + ca + 2], sAff[cpad + ca + 3]);
                const bool cok = (i == G - 1 && has_last) ? last_ok : true;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const bool ok = myNbr[k * ROWS + 16 * rt] >= 0 && cok;
                    float4 x = av[rt];
                    x.x = fmaf(x.x, sc.x, sh.x); x.y = fmaf(x.y, sc.y, sh.y); x.z = fmaf(x.z, sc.z, sh.z); x.w = fmaf(x.w, sc.w, sh.w);
                    if constexpr (MODE == 2 || MODE == 4) {      // (no branch: v_max + v_bfi per value; a NaN stays a NaN without the ReLU)
                        x.x = relu_sel(x.x, relu_mask); x.y = relu_sel(x.y, relu_mask);
                        x.z = relu_sel(x.z, relu_mask); x.w = relu_sel(x.w, relu_mask);
                    } else {
                        if (p.in_relu) { x.x = relu_nc(x.x); x.y = relu_nc(x.y); x.z = relu_nc(x.z); x.w = relu_nc(x.w); }
                        if (p.Cin & 3) {
                            const int c = ca;
                            if (c + 1 >= p.Cin) x.y = 0.0f;
                            if (c + 2 >= p.Cin) x.z = 0.0f;
                            if (c + 3 >= p.Cin) x.w = 0.0f;
                        }
                    }
                    av[rt] = ok ? x : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
            if constexpr (BF) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (t8) av[rt].z = av[rt].w = 0.0f;     // (the fp32 form skips these two k slots; their weights are zeros)
                    qa[i][rt] = bf_split(av[rt]);
                }
#pragma unroll
                for (int t = 0; t < CTM; ++t) qb[i][t] = bf_split(g.b[i][t]);
                continue;
            }
            // independent accumulators alternate: a 16x16x4 MFMA issues every 32 cycles and returns after 40
#define EP_DIRECT_STEP(comp, set)                                                                                                    \
    do {                                                                                                                             \
        if (EP_DIRECT_ABL & 16) {                                                                                                    \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                                                      \
                _Pragma("unroll") for (int t = 0; t < CTM; ++t) acc[0][rt][t][0] += av[rt].comp + g.b[i][t].comp;                    \
                if constexpr (TAIL) acct[rt][0][0] += av[rt].comp + g.bt[i][0].comp + g.bt[i][1].comp;                               \
            }                                                                                                                        \
            break;                                                                                                                   \
        }                                                                                                                            \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                            \
            _Pragma("unroll") for (int t = 0; t < CTM; ++t)                                                                          \
                acc[(set) % NS][rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].comp, g.b[i][t].comp, acc[(set) % NS][rt][t], 0, 0, 0); \
        if constexpr (TAIL) {                                                                                                        \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                        \
                _Pragma("unroll") for (int cg = 0; cg < 2; ++cg)                                                                     \
                    acct[rt][cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[rt].comp, g.bt[i][cg].comp, acct[rt][cg], 0, 0, 0);         \
        }                                                                                                                            \
    } while (0)
            EP_DIRECT_STEP(x, 0);
            EP_DIRECT_STEP(y, 1);
            if (!t8) {
                EP_DIRECT_STEP(z, 0);
                EP_DIRECT_STEP(w, 1);
            }
#undef EP_DIRECT_STEP
        }
        if constexpr (BF) {     // chunks in pairs on the K = 32 instruction, an odd last one on K = 16; three products each
#pragma unroll
            for (int i = 0; i + 1 < G; i += 2)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < CTM; ++t) {
                        f32x4 c = acc[0][rt][t];
                        c = bf_mfma32(qa[i][rt].lo, qa[i + 1][rt].lo, qb[i][t].hi, qb[i + 1][t].hi, c);    // (small terms first)
                        c = bf_mfma32(qa[i][rt].hi, qa[i + 1][rt].hi, qb[i][t].lo, qb[i + 1][t].lo, c);
                        c = bf_mfma32(qa[i][rt].hi, qa[i + 1][rt].hi, qb[i][t].hi, qb[i + 1][t].hi, c);
                        acc[0][rt][t] = c;
                    }
            if constexpr (G & 1) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < CTM; ++t) {
                        f32x4 c = acc[0][rt][t];
                        c = bf_mfma16(qa[G - 1][rt].lo, qb[G - 1][t].hi, c);
                        c = bf_mfma16(qa[G - 1][rt].hi, qb[G - 1][t].lo, c);
                        c = bf_mfma16(qa[G - 1][rt].hi, qb[G - 1][t].hi, c);
                        acc[0][rt][t] = c;
                    }
            }
        }
    };
    // Main loop.  The loads of stage u + 1 are in flight while stage u runs its MFMAs.  MODE 0: a scheduling fence between the two,
    // i.e. a burst of <= 15 loads, then <= 72 MFMAs.  MODE 1 / 2 (the consume step is one basic block): no fence — the loads are
    // spread among the MFMAs, one VMEM read per kMixMfma matrix instructions (sched_group_barrier), so a wave keeps feeding the
    // matrix pipe while its loads issue, and the registers of the stage in flight fill as they are needed (100 + 40 instead of
    // 144 + 24 on 48 -> 24): 5-8 % on the long lists (profiles/r06/conv_interleave_ab.txt).  An odd last stage is peeled off so
    // that the paired body holds no branch.  (One kernel holding both forms of the loop needs 236 + 44 registers: they are
    // separate instantiations.)
    if constexpr (MODE == 0) {
        if (!(p.debug & 1) && U > 0) {
            Stage s_a, s_b;
            LiveCursor cf(live), cc(live);     // the fetches run one to two stages ahead of the MFMAs
            fetch(cf, s_a);
            for (int u = 0; u < U; u += 2) {
                cf.next(PARTS);
                fetch(cf, s_b);
                __builtin_amdgcn_sched_barrier(0);
                consume(cc, s_a);
                cc.next(PARTS);
                __builtin_amdgcn_sched_barrier(0);
                cf.next(PARTS);
                fetch(cf, s_a);
                __builtin_amdgcn_sched_barrier(0);
                if (u + 1 < U) consume(cc, s_b);
                cc.next(PARTS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (U > 0) {
        constexpr int kStageLoads = RT * G + G * CTM + (TAIL ? 2 * G : 0);
        constexpr int kStageMfma = BF ? 3 * RT * CTM * ((G + 1) / 2) : (G * 4 - (MODE >= 3 ? 2 : 0)) * (RT * CTM + (TAIL ? 2 * RT : 0));
        constexpr int kMixMfma = (kStageMfma + kStageLoads - 1) / kStageLoads + EP_MIX_EXTRA;
        Stage s_a, s_b;
        LiveCursor cf(live), cc(live);
        fetch(cf, s_a);
        auto mix = [&]() {
#pragma unroll
            for (int sg = 0; sg < kStageLoads; ++sg) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, kMixMfma, 0);
            }
        };
        int u = 0;
        for (; u + 1 < U; u += 2) {
            cf.next(PARTS);
            fetch(cf, s_b);
            consume(cc, s_a);
            mix();
            cc.next(PARTS);
            __builtin_amdgcn_sched_barrier(0);
            cf.next(PARTS);
            fetch(cf, s_a);
            consume(cc, s_b);
            mix();
            cc.next(PARTS);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (u < U) consume(cc, s_a);
    }
    if constexpr (NS == 2) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < CTM; ++t) acc[0][rt][t] += acc[1][rt][t];
    }
    if constexpr (TAIL) {
        f32x4 full[RT][CT];
        f32x4 last[RT];
        tail_to_tile(acct, last, lane);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int t = 0; t < CTM; ++t) full[rt][t] = acc[0][rt][t];
            full[rt][CT - 1] = last[rt];
        }
        direct_epilogue<CT>(p, full, row0, sStat);
    } else {
        direct_epilogue<CT>(p, acc[0], row0, sStat);
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent form for long lists whose packed weights fit the LDS (round 6).  What the ablations of the kernel above say about
// the cfg4-leading layer (48 -> 24 on 320,868 rows, profiles/r06/conv_direct_ablate.txt): without its MFMAs the launch still
// takes 72 % of its time, without its weight loads 13 % less, without its gathers 13 % less — every wave re-reads the SAME
// operand-order weights through the vector L1 (9 of its 15 16-byte loads per offset; 64 B/clk/CU), and a wave stuck issuing
// loads cannot issue MFMAs.  Here ONE workgroup of 8 waves per CU copies the whole packing into LDS once (124 KB for 48 -> 24;
// ds_read_b128 delivers 256 B/clk/CU, no L1 traffic), and its waves then pull 32-row jobs from a device counter until the list
// is exhausted: no barrier after the prologue, no per-workgroup launch / staging cost, the dispatch balanced at the granularity
// of a wave's job whatever else runs on the chip.  The map slice of a job is staged by the wave itself in its own LDS window
// (the ballots of the staging passes are its live-offset mask); gathers, MFMA loop and epilogue are the direct kernel's; the
// BatchNorm summaries are per job (32 rows; the caller sizes them through eprecon_conv_desc_partial_rows).
// Job counters: same-address device atomics retire one per ~21 ns on this part (tools/probes/atomic_probe.hip), so ONE counter
// for the launch's ~12k pulls would cost more than the convolution (measured: 278 us).  The jobs are split into 8 contiguous
// shares — workgroup b pulls from share b % 8, the XCD the dispatcher puts it on when the chip is free; nothing depends on that —
// each with its own pair (next job, waves done) 64 bytes apart, out of a zero-initialised device array, one set per launch in
// rotation.  Every wave's LAST pull fails by construction and the last wave of a share to leave puts its pair back to zero, so a
// set can be reused (HIP graph replays reuse theirs).
// ---------------------------------------------------------------------------------------------
constexpr int kPersistWaves = 8;
constexpr int kJobRows = 16 * kRT;
constexpr int kPersistSlots = 512, kShares = 8, kCtrStride = 16;     // (16 uints = 64 B between the pairs of a set)
__device__ unsigned int g_persist_ctr[kPersistSlots * kShares * kCtrStride];

template <int CT, int KCH, bool TAIL, int G = stage_chunks(KCH)>
__global__ __launch_bounds__(64 * kPersistWaves) void spconv_persist16_kernel(ConvParams p, int njobs, unsigned int *ctr)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RT = kRT, ROWS = kJobRows;
    constexpr int CTM = TAIL ? CT - 1 : CT;
    constexpr int CTA = CTM > 0 ? CTM : 1;
    constexpr int PARTS = KCH / G;
    constexpr int cpad = 16 * KCH;
    constexpr int kQuads = CTM * 64 + (TAIL ? 32 : 0);      // float4s of one (offset, chunk) block in LDS
    float4 *sW = reinterpret_cast<float4 *>(smem);                                  // [K][KCH][kQuads]
    int *sNbr = reinterpret_cast<int *>(sW + (size_t)p.K * KCH * kQuads);           // [waves][K][32]
    float *sAff = reinterpret_cast<float *>(sNbr + kPersistWaves * p.K * ROWS);     // [2][cpad]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, q = lane >> 4;

    {   // the packing -> LDS, once per CU
        const float4 *wm = reinterpret_cast<const float4 *>(p.wq16);
        const float4 *wt = wm + (size_t)p.K * KCH * CT * 64;
        const int blocks = p.K * KCH;
        for (int e = tid; e < blocks * kQuads; e += 64 * kPersistWaves) {
            const int blk = e / kQuads, r = e - blk * kQuads;
            sW[e] = r < CTM * 64 ? wm[(size_t)blk * CT * 64 + r] : wt[(size_t)blk * 32 + (r - CTM * 64)];
        }
    }
    stage_in_affine<64 * kPersistWaves>(p, sAff, cpad, tid);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const unsigned row_bytes = (unsigned)p.ld_x * 4u, oob = (unsigned)p.x_bytes;
    int *wNbr = sNbr + wave * p.K * ROWS;
    const int *myNbr = wNbr + l16;
    const int last_c = p.Cin - 16 * (KCH - 1);
    const bool tail8 = last_c <= 8;
    const unsigned cq = 16u * (unsigned)q;
    const unsigned cq_last = tail8 ? 8u * (unsigned)q : cq;
    const bool last_ok = (tail8 ? 2 : 4) * q < last_c;
    const int tq = CTM * 64 + 4 * q + (lane & 3);       // this lane's float4 of a block's tail section (+ 16 cg)
    const int total = p.K * ROWS;

    struct Stage {
        float4 a[G][RT];
        float4 b[G][CTA];
        float4 bt[G][TAIL ? 2 : 1];
    };

    // this workgroup's share of the jobs: [job_lo, job_hi)
    const unsigned share = blockIdx.x % (unsigned)kShares;
    ctr += share * kCtrStride;
    const unsigned per = ((unsigned)njobs + kShares - 1) / kShares;
    const unsigned job_lo = share * per, job_hi = min(job_lo + per, (unsigned)njobs);
    unsigned job = 0;
    if (lane == 0) job = atomicAdd(&ctr[0], 1u);
    job = (unsigned)__builtin_amdgcn_readfirstlane((int)job) + job_lo;
    while (job < job_hi) {
        const int row0 = (int)job * ROWS;
        // the job's slice of the kernel map -> the wave's LDS window [K][32]; pass `it` covers offsets 2 it (lanes 0 .. 31) and
        // 2 it + 1: the halves of its ballot are those offsets' liveness
        unsigned live = 0u;
        {
            int jv[kMapLoads];
#pragma unroll
            for (int it = 0; it < kMapLoads; ++it) {
                const int e = lane + 64 * it;
                const int k = e >> 5, row = row0 + (e & 31);
                int j = -1;
                if (e < total && row < p.n_out) j = p.nbr ? p.nbr[(size_t)k * p.n_out + row] : row;
                jv[it] = j;
            }
#pragma unroll
            for (int it = 0; it < kMapLoads; ++it) {
                const int e = lane + 64 * it;
                if (e < total) wNbr[e] = jv[it];
                const unsigned long long b = __ballot(jv[it] >= 0);
                live |= ((unsigned)b != 0u ? 1u : 0u) << (2 * it);
                live |= ((unsigned)(b >> 32) != 0u ? 1u : 0u) << (2 * it + 1);
            }
        }
        live = (unsigned)__builtin_amdgcn_readfirstlane((int)live);
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(&ctr[0], 1u);        // the next job's index travels while this one runs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the window is written before any lane reads another lane's entry

        constexpr int NS = CTM == 1 ? 2 : 1;       // (two accumulator sets when a row tile has one full column tile: see above)
        f32x4 acc[NS][RT][CTA];
        f32x4 acct[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)
#pragma unroll
                for (int t = 0; t < CTA; ++t) acc[ns][rt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            acct[rt][0] = acct[rt][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        const int U = __builtin_popcount(live) * PARTS;

        auto fetch = [&](const LiveCursor &c, Stage &g) {
            const int k = c.k, part = c.part;
            const unsigned xs = 64u * (unsigned)(part * G);
            const bool has_last = part == PARTS - 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int j = myNbr[k * ROWS + 16 * rt];
                const unsigned rowsel = j >= 0 ? __umul24((unsigned)j, row_bytes) : oob;
                const unsigned v0 = rowsel + cq;
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    unsigned off = v0 + 64u * i;
                    if (i == G - 1 && has_last) off = last_ok ? rowsel + cq_last + 64u * i : oob;
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, xs, 0);
                    g.a[i][rt] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            }
            const float4 *wb = sW + (size_t)(k * KCH + part * G) * kQuads;
#pragma unroll
            for (int i = 0; i < G; ++i) {
#pragma unroll
                for (int t = 0; t < CTM; ++t) g.b[i][t] = wb[i * kQuads + t * 64 + lane];
                if constexpr (TAIL) {
                    g.bt[i][0] = wb[i * kQuads + tq];
                    g.bt[i][1] = wb[i * kQuads + tq + 16];
                }
            }
        };
        auto consume = [&](const LiveCursor &c, const Stage &g) {
            const int k = c.k, part = c.part;
            const bool has_last = part == PARTS - 1;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const bool t8 = tail8 && i == G - 1 && has_last;
                float4 av[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) av[rt] = g.a[i][rt];
                if ((p.Cin & 3) && i == G - 1 && has_last) {
                    const int c = 16 * (KCH - 1) + (t8 ? 2 : 4) * q;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        if (c + 1 >= p.Cin) av[rt].y = 0.0f;
                        if (c + 2 >= p.Cin) av[rt].z = 0.0f;
                        if (c + 3 >= p.Cin) av[rt].w = 0.0f;
                    }
                }
                if (p.in_scale) {
                    const int kc = part * G + i;
                    const int ca = 16 * kc + (t8 ? 2 : 4) * q;
                    const float4 sc = make_float4(sAff[ca], sAff[ca + 1], sAff[ca + 2], sAff[ca + 3]);
                    const float4 sh = make_float4(sAff[cpad + ca], sAff[cpad + ca + 1], sAff[cpad + ca + 2], sAff[cpad + ca + 3]);
                    const bool cok = (i == G - 1 && has_last) ? last_ok : true;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const bool ok = myNbr[k * ROWS + 16 * rt] >= 0 && cok;
                        float4 x = av[rt];
                        x.x = fmaf(x.x, sc.x, sh.x); x.y = fmaf(x.y, sc.y, sh.y); x.z = fmaf(x.z, sc.z, sh.z); x.w = fmaf(x.w, sc.w, sh.w);
                        if (p.in_relu) { x.x = relu_nc(x.x); x.y = relu_nc(x.y); x.z = relu_nc(x.z); x.w = relu_nc(x.w); }
                        if (p.Cin & 3) {
                            if (ca + 1 >= p.Cin) x.y = 0.0f;
                            if (ca + 2 >= p.Cin) x.z = 0.0f;
                            if (ca + 3 >= p.Cin) x.w = 0.0f;
                        }
                        av[rt] = ok ? x : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
                }
#define EP_PERSIST_STEP(comp, set)                                                                                                   \
    do {                                                                                                                             \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                            \
            _Pragma("unroll") for (int t = 0; t < CTM; ++t)                                                                          \
                acc[(set) % NS][rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].comp, g.b[i][t].comp, acc[(set) % NS][rt][t], 0, 0, 0); \
        if constexpr (TAIL) {                                                                                                        \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                        \
                _Pragma("unroll") for (int cg = 0; cg < 2; ++cg)                                                                     \
                    acct[rt][cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[rt].comp, g.bt[i][cg].comp, acct[rt][cg], 0, 0, 0);         \
        }                                                                                                                            \
    } while (0)
                EP_PERSIST_STEP(x, 0);
                EP_PERSIST_STEP(y, 1);
                if (!t8) {
                    EP_PERSIST_STEP(z, 0);
                    EP_PERSIST_STEP(w, 1);
                }
#undef EP_PERSIST_STEP
            }
        };
        if (U > 0) {
            Stage s_a, s_b;
            LiveCursor cf(live), cc(live);
            fetch(cf, s_a);
            for (int u = 0; u < U; u += 2) {
                cf.next(PARTS);
                fetch(cf, s_b);
                __builtin_amdgcn_sched_barrier(0);
                consume(cc, s_a);
                cc.next(PARTS);
                __builtin_amdgcn_sched_barrier(0);
                cf.next(PARTS);
                fetch(cf, s_a);
                __builtin_amdgcn_sched_barrier(0);
                if (u + 1 < U) consume(cc, s_b);
                cc.next(PARTS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (NS == 2) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < CTM; ++t) acc[0][rt][t] += acc[1][rt][t];
        }
        if constexpr (TAIL) {
            f32x4 full[RT][CT];
            f32x4 last[RT];
            tail_to_tile(acct, last, lane);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int t = 0; t < CTM; ++t) full[rt][t] = acc[0][rt][t];
                full[rt][CT - 1] = last[rt];
            }
            direct_epilogue<CT, true>(p, full, row0, nullptr, (int)job);
        } else {
            direct_epilogue<CT, true>(p, acc[0], row0, nullptr, (int)job);
        }
        job = (unsigned)__builtin_amdgcn_readfirstlane((int)next) + job_lo;
    }
    if (lane == 0) {
        // workgroups of this share: b = share, share + 8, ... < gridDim.x
        const unsigned mates = (gridDim.x - share + kShares - 1) / kShares;
        const unsigned done = atomicAdd(&ctr[1], 1u);
        if (done == mates * (unsigned)kPersistWaves - 1u) {      // every wave of the share has made its last (failing) pull
            __hip_atomic_store(&ctr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctr[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Any chunk count (C_in > 96): the (offset, chunk) sequence is walked with run-time indices, two chunks per stage.
template <int CT>
__global__ __launch_bounds__(256) void spconv_direct16_generic_kernel(ConvParams p, int kch)
{
    constexpr int RT = kRT, G = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 64 * RT;
    constexpr int NR = 4 * RT;
    int *sNbr = reinterpret_cast<int *>(smem);                        // [K][ROWS]
    float *sStat = reinterpret_cast<float *>(sNbr + p.K * ROWS);      // [4 waves][3][16 CT]
    const int cpad = 16 * kch;
    float *sAff = sStat + kWaves * 3 * 16 * CT;                       // [2][cpad]
    int *sFlag = reinterpret_cast<int *>(sAff + 2 * cpad);            // [K][4 waves] the wave has a neighbour at the offset
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, q = lane >> 4;
    const int row0 = (int)blockIdx.x * ROWS;

    stage_in_affine<256>(p, sAff, cpad, tid);
    // (live offsets of the wave's 32 rows: see the template kernel)
    const unsigned live = stage_map(p, row0, sNbr, sFlag, tid);

    constexpr int NS = CT == 1 ? 2 : 1;       // (two accumulator sets for a single column tile: see the template kernel)
    f32x4 acc[NS][RT][CT];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[ns][rt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const unsigned row_bytes = (unsigned)p.ld_x * 4u, oob = (unsigned)p.x_bytes;
    const int S = __builtin_popcount(live) * kch;  // (live offset, chunk) steps
    const unsigned step_bytes = (unsigned)CT * 1024u;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wq16), 0, (int)((unsigned)p.K * kch * step_bytes), 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    const int *myNbr = sNbr + wave * 16 * RT + l16;
    const unsigned cq = 16u * (unsigned)q;         // byte offset of this lane's four channels inside a chunk
    // a last chunk of <= 8 channels (C_in = 8, 24, 40): lane group q takes channels 2 q, 2 q + 1 and the chunk is two MFMAs
    const bool tail8 = p.Cin - 16 * (kch - 1) <= 8;

    struct Stage {
        float4 a[G][RT];
        float4 b[G][CT];
        unsigned live;      // bit (i * RT + rt): the neighbour of step i, row tile rt exists (only read with in_scale)
    };
    // the G steps at the cursor, which moves on (and stays on the last step: a step fetched past the end is not used)
    auto fetch = [&](LiveCursor &c, Stage &g) {
        g.live = 0u;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int k = c.k, kc = c.part;
            c.next(kch);
            const bool t8 = tail8 && kc == kch - 1;                         // (uniform)
            const unsigned cbytes = 64u * (unsigned)kc + (t8 ? cq >> 1 : cq);
            const bool cok = 16 * kc + (t8 ? 2 : 4) * q < p.Cin;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int j = myNbr[k * ROWS + 16 * rt];
                const bool ok = j >= 0 && cok;
                const unsigned off = ok ? __umul24((unsigned)j, row_bytes) + cbytes : oob;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
                g.a[i][rt] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                g.live |= (ok ? 1u : 0u) << (i * RT + rt);
            }
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + (unsigned)t * 1024u, (unsigned)(k * kch + kc) * step_bytes, 0);
                g.b[i][t] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        }
    };
    auto consume = [&](int s0, const Stage &g) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (s0 + i < S) {      // (uniform)
                float4 av[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) av[rt] = g.a[i][rt];
                const int kc = (s0 + i) % kch;
                const bool t8 = tail8 && kc == kch - 1;      // (uniform) .z / .w of the gathered values are not used
                if (p.Cin & 3) {   // (uniform) ragged channel count on a padded pitch: whatever sits in the pad lanes stays out
                    const int c = 16 * kc + (t8 ? 2 : 4) * q;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        if (c + 1 >= p.Cin) av[rt].y = 0.0f;
                        if (c + 2 >= p.Cin) av[rt].z = 0.0f;
                        if (c + 3 >= p.Cin) av[rt].w = 0.0f;
                    }
                }
                if (p.in_scale) {  // (uniform) the producer's pending BatchNorm (+ ReLU) on the gathered values
                    const int ca = 16 * kc + (t8 ? 2 : 4) * q;   // (8-byte aligned in the tail form: read as scalars)
                    const float4 sc = make_float4(sAff[ca], sAff[ca + 1], sAff[ca + 2], sAff[ca + 3]);
                    const float4 sh = make_float4(sAff[cpad + ca], sAff[cpad + ca + 1], sAff[cpad + ca + 2], sAff[cpad + ca + 3]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const bool ok = (g.live >> (i * RT + rt)) & 1u;
                        float4 x = av[rt];
                        x.x = fmaf(x.x, sc.x, sh.x); x.y = fmaf(x.y, sc.y, sh.y); x.z = fmaf(x.z, sc.z, sh.z); x.w = fmaf(x.w, sc.w, sh.w);
                        if (p.in_relu) { x.x = relu_nc(x.x); x.y = relu_nc(x.y); x.z = relu_nc(x.z); x.w = relu_nc(x.w); }
                        av[rt] = ok ? x : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
                }
                // independent accumulators alternate: a 16x16x4 MFMA issues every 32 cycles and returns after 40
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < CT; ++t) acc[0 % NS][rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].x, g.b[i][t].x, acc[0 % NS][rt][t], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < CT; ++t) acc[1 % NS][rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].y, g.b[i][t].y, acc[1 % NS][rt][t], 0, 0, 0);
                if (!t8) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int t = 0; t < CT; ++t) acc[0 % NS][rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].z, g.b[i][t].z, acc[0 % NS][rt][t], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int t = 0; t < CT; ++t) acc[1 % NS][rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].w, g.b[i][t].w, acc[1 % NS][rt][t], 0, 0, 0);
                }
            }
        }
    };
    if (!(p.debug & 1) && S > 0) {
        Stage s_a, s_b;
        LiveCursor cf(live);
        fetch(cf, s_a);
        for (int s0 = 0; s0 < S; s0 += 2 * G) {
            fetch(cf, s_b);
            __builtin_amdgcn_sched_barrier(0);
            consume(s0, s_a);
            __builtin_amdgcn_sched_barrier(0);
            fetch(cf, s_a);
            __builtin_amdgcn_sched_barrier(0);
            consume(s0 + G, s_b);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (NS == 2) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[0][rt][0] += acc[1][rt][0];
    }
    direct_epilogue<CT>(p, acc[0], row0, sStat);
}


// EPRECON_CONV_TAIL8=0: C_out = 16 m + 8 on padded 16-column tiles (the round-5 form; read per launch: tests flip it)
static bool tail8_enabled()
{
    const char *e = getenv("EPRECON_CONV_TAIL8");
    return !(e && e[0] == '0');
}

// EPRECON_CONV_BF16X3=1: the template kernel's bf16x3 operand form (C_in <= 96, C_out <= 64; read per launch).  OFF by default:
// the library's results are exact-fp32 products; this is the opt-in with a stated error budget.
static bool bf16x3_enabled()
{
    const char *e = getenv("EPRECON_CONV_BF16X3");
    return e && e[0] == '1';
}

// EPRECON_CONV_INTERLEAVE=0: the fenced schedule (loads of a stage, then the previous stage's MFMAs) for every launch of the template
// kernel (read per launch); default: branch-free layers spread the loads among the MFMAs.  Same arithmetic in the same order.
static bool interleave_enabled()
{
    const char *e = getenv("EPRECON_CONV_INTERLEAVE");
    return !(e && e[0] == '0');
}

// EPRECON_CONV_STAGE_DEPTH=0: always the deepest prefetch stage (the round-5 rule; read per launch)
static bool stage_depth_enabled()
{
    const char *e = getenv("EPRECON_CONV_STAGE_DEPTH");
    return !(e && e[0] == '0');
}

constexpr size_t kLdsBytes = 160 * 1024;

static size_t persist_lds_bytes(const ConvParams &p)
{
    const int kch = (p.Cin + 15) / 16, ct = (p.Cout + 15) / 16;
    const bool tail = p.Cout - 16 * (ct - 1) <= 8;
    const size_t quads = (size_t)(tail ? ct - 1 : ct) * 64 + (tail ? 32 : 0);
    return (size_t)p.K * kch * quads * 16 + (size_t)kPersistWaves * p.K * kJobRows * sizeof(int) + (size_t)2 * 16 * kch * sizeof(float);
}

static int device_cus()
{
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// EPRECON_CONV_PERSIST=1: long lists with LDS-sized weights on the persistent kernel (read per launch).  OFF by default: measured
// equal to the per-tile kernel on the cfg4-leading layer (48 -> 24 on 320,868 rows: 203.6 against 203.3 us) and 10-19 % slower
// on the 198k-row layers, whose 24.2 jobs per CU are 2.02 per wave of three resident ones — a third round for 1 % of the jobs
// (profiles/r06/conv_forms_ab.txt).
static bool persist_enabled()
{
    const char *e = getenv("EPRECON_CONV_PERSIST");
    return e && e[0] == '1';
}

// (opt-in: any list with a job per wave of a few workgroups may take it)
constexpr int kPersistMinRows = 8192;

static bool persist_ok(const ConvParams &p)
{
    if (!persist_enabled() || (p.Cin + 15) / 16 > 6) return false;
    if (p.Cout - 16 * ((p.Cout + 15) / 16 - 1) <= 8 && !tail8_enabled()) return false;    // (its 8-column tail is the 4x4x1 form)
    if (p.bn_partial && !p.flex_partial) return false;          // its summaries are per 32-row job
    if (persist_lds_bytes(p) > kLdsBytes) return false;
    {   // (the instantiated forms: launch_k's rule)
        const int kch = (p.Cin + 15) / 16, ct = (p.Cout + 15) / 16;
        const bool tail = p.Cout - 16 * (ct - 1) <= 8;
        if (kch * ((tail ? ct - 1 : ct) * 1024 + (tail ? 512 : 0)) > 4864) return false;
    }
    return p.n_out >= kPersistMinRows;
}

template <int CT, int KCH, bool TAIL>
int launch_persist(const ConvParams &p, hipStream_t st)
{
    static bool attr_set = false;
    auto kern = spconv_persist16_kernel<CT, KCH, TAIL>;
    if (!attr_set) {
        EP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        attr_set = true;
    }
    static unsigned int *ctr_base = nullptr;
    if (!ctr_base) EP_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void **>(&ctr_base), HIP_SYMBOL(g_persist_ctr)));
    static std::atomic<unsigned> slot{0};
    unsigned int *ctr = ctr_base + (size_t)kShares * kCtrStride * (slot.fetch_add(1u) % (unsigned)kPersistSlots);
    const int njobs = ceil_div(p.n_out, kJobRows);
    const int grid = max(kShares, min(device_cus(), ceil_div(njobs, kPersistWaves)));    // (every share has a workgroup)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kPersistWaves), persist_lds_bytes(p), st, p, njobs, ctr);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

template <int CT, int KCH>
int launch_k(const ConvParams &p_in, hipStream_t st)
{
    const ConvParams &p = p_in;
    const size_t lds = (size_t)p.K * kDirectRows * sizeof(int) + (size_t)kWaves * 3 * 16 * CT * sizeof(float) + (size_t)2 * 16 * KCH * sizeof(float) + (size_t)p.K * kWaves * sizeof(int);
    // (one chunk per stage — 80 registers, six waves per SIMD instead of three — measured no faster: 266 vs 250 us on 48 -> 24)
    const int rem = p.Cout - 16 * (CT - 1);       // columns of the last tile
    // (instantiated only where a 27-offset packing can fit the LDS beside the eight job windows: <= 4.9 KB per offset)
    if constexpr (KCH * (CT * 1024 - 512) <= 4864) {
        if (persist_ok(p)) {
            if (rem <= 8) return launch_persist<CT, KCH, true>(p, st);
            if constexpr (KCH * CT * 1024 <= 4864) return launch_persist<CT, KCH, false>(p, st);
        }
    }
    const dim3 grid((unsigned)ceil_div(p.n_out, kDirectRows));
    if (bf16x3_enabled()) {
        const int bmode = (interleave_enabled() && p.Cin - 16 * (KCH - 1) > 8 && (p.Cin & 3) == 0) ? (p.in_scale ? 2 : 1) : 0;
        if (bmode == 2) hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, stage_chunks(KCH), true, 2>), grid, dim3(256), lds, st, p);
        else if (bmode == 1) hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, stage_chunks(KCH), true, 1>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, stage_chunks(KCH), true>), grid, dim3(256), lds, st, p);
        EP_LAUNCH_CHECK();
        return EPRECON_OK;
    }
    const bool tail = rem <= 8 && tail8_enabled();
    constexpr int G0 = stage_chunks(KCH);
    // branch-free layers (no 8-channel or ragged last chunk) take the interleaved instantiations: 1 without, 2 with a pending
    // BatchNorm of the input (profiles/r06/conv_interleave_ab.txt: -10 ... -12 % on 48 -> 24 / 80 -> 40 / 32 -> 24 / 48 -> 48 without,
    // -5 ... -18 % with; where the stage-depth rule below finds room for one more workgroup per CU, the layers WITH a pending
    // BatchNorm stay on its shallow fenced form: 48 -> 48 on 93,513 rows 132.7 against 140.0 us)
    int mode = (interleave_enabled() && p.Cin - 16 * (KCH - 1) > 8 && (p.Cin & 3) == 0) ? (p.in_scale ? 2 : 1) : 0;
    if constexpr (G0 == KCH) {      // (C_in = 8, 24, 40: the 8-channel last chunk is part of every stage)
        if (interleave_enabled() && p.Cin - 16 * (KCH - 1) <= 8 && (p.Cin & 3) == 0) mode = p.in_scale ? 4 : 3;
    }
    // Medium lists (a few workgroups per CU): the kernel's registers allow two workgroups per CU for the wide layers (96 -> 48,
    // 48 -> 48: three chunks per stage), so 583 workgroups (74,568 rows) run as one full wave of 512 and a second one that is
    // 14 % full.  The same kernel with FEWER chunks per stage (kAltG: fewer prefetch registers, one more workgroup per CU) is
    // ~6 % slower per workgroup and holds them all at once: 96 -> 48 on 74,568 rows 246 -> 199 us, 48 -> 48 on 93,513 rows
    // 139 -> 116 us (profiles/r06/conv_stage_depth_ab.txt).  Chosen per launch from the two forms' occupancies.
    constexpr int kAltG = KCH == 6 ? 2 : (KCH == 3 ? 1 : 0);
    if constexpr (kAltG != 0) {
        static int occ[2][2] = {{0, 0}, {0, 0}};      // [tail][primary, alternative] workgroups per CU
        int *o = occ[tail ? 1 : 0];
        if (!o[0]) {
            const void *kp = tail ? reinterpret_cast<const void *>(spconv_direct16_kernel<CT, KCH, true, G0>)
                                  : reinterpret_cast<const void *>(spconv_direct16_kernel<CT, KCH, false, G0>);
            const void *ka = tail ? reinterpret_cast<const void *>(spconv_direct16_kernel<CT, KCH, true, kAltG>)
                                  : reinterpret_cast<const void *>(spconv_direct16_kernel<CT, KCH, false, kAltG>);
            int a = 0, b = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, kp, 256, lds) != hipSuccess || a <= 0) a = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ka, 256, lds) != hipSuccess || b <= 0) b = a;
            o[1] = b;
            o[0] = a;
        }
        const int64_t wgs = grid.x, cus = device_cus();
        // waves of workgroups x workgroups sharing a SIMD: what a launch costs in units of one workgroup running alone
        const double cost_p = (double)ceil_div(wgs, cus * o[0]) * o[0];
        const double cost_a = (double)ceil_div(wgs, cus * o[1]) * o[1] * 1.06;
        if (stage_depth_enabled() && o[1] > o[0] && cost_a < 0.85 * cost_p && mode != 1 && mode != 3) {
            if (tail) hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, true, kAltG>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, kAltG>), grid, dim3(256), lds, st, p);
            EP_LAUNCH_CHECK();
            return EPRECON_OK;
        }
    }
    if constexpr (G0 == KCH) {
        if (mode == 4) {
            if (tail) hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, true, G0, false, 4>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, G0, false, 4>), grid, dim3(256), lds, st, p);
            EP_LAUNCH_CHECK();
            return EPRECON_OK;
        }
        if (mode == 3) {
            if (tail) hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, true, G0, false, 3>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, G0, false, 3>), grid, dim3(256), lds, st, p);
            EP_LAUNCH_CHECK();
            return EPRECON_OK;
        }
    }
    if (mode == 2) {
        if (tail) hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, true, G0, false, 2>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, G0, false, 2>), grid, dim3(256), lds, st, p);
    } else if (mode == 1) {
        if (tail) hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, true, G0, false, 1>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, G0, false, 1>), grid, dim3(256), lds, st, p);
    } else if (tail) {
        hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, true, G0>), grid, dim3(256), lds, st, p);
    } else {
        hipLaunchKernelGGL((spconv_direct16_kernel<CT, KCH, false, G0>), grid, dim3(256), lds, st, p);
    }
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

template <int CT>
int launch_ct(const ConvParams &p, hipStream_t st)
{
    const int kch = (p.Cin + 15) / 16;
    switch (kch) {
        case 1: return launch_k<CT, 1>(p, st);
        case 2: return launch_k<CT, 2>(p, st);
        case 3: return launch_k<CT, 3>(p, st);
        case 4: return launch_k<CT, 4>(p, st);
        case 5: return launch_k<CT, 5>(p, st);
        case 6: return launch_k<CT, 6>(p, st);
        default: break;
    }
    const size_t lds = (size_t)p.K * kDirectRows * sizeof(int) + (size_t)kWaves * 3 * 16 * CT * sizeof(float) + (size_t)2 * 16 * kch * sizeof(float) + (size_t)p.K * kWaves * sizeof(int);
    hipLaunchKernelGGL((spconv_direct16_generic_kernel<CT>), dim3((unsigned)ceil_div(p.n_out, kDirectRows)), dim3(256), lds, st, p, kch);
    EP_LAUNCH_CHECK();
    return EPRECON_OK;
}

}  // namespace
}  // namespace epconv
