// Shared between the translation units of the convolution family (sparse_conv.hip, sparse_conv_direct.hip): the launch
// parameters, the Chan merge of BatchNorm summaries and the entry points of the direct gather kernel.
#pragma once
#include "common.hpp"

namespace epconv {

struct ConvParams {
    const float *x;      // [n_in, ld_x] (+ x_col0 folded into the pointer)
    const int32_t *nbr;  // [K][n_out] or nullptr (K must be 1: identity map)
    const float *w;      // [K][Cin][Cout]
    const float *bias;   // [Cout] or nullptr
    float *out;          // [n_out, ld_out]
    int n_out, K, Cin, Cout, ld_x, ld_out;
    int64_t x_bytes;     // bytes addressable from x (buffer-load gathers: rows past it read as zeros)
    int relu;            // fused ReLU epilogue
    int accumulate;      // out += result instead of out = result
    const float *res;    // optional [n_out, ld_res]: added after the ReLU (x + ReLU(conv(x)) blocks)
    int ld_res;
    float *bn_partial;   // optional [gridDim.x][3][Cout]: per-workgroup (count, mean, M2) of the stored values
    // BatchNorm of the INPUT applied while gathering: a = [relu](x * in_scale[c] + in_shift[c]) (nullptr: a = x)
    const float *in_scale, *in_shift;
    int in_relu;
    // the same for the residual operand (columns of the output)
    const float *res_scale, *res_shift;
    int res_relu;
    // LayerNorm over the Cout channels of every output row, after bias / ReLU / residual (needs all
    // columns in one workgroup): out = [relu]( LN(v) * ln_gamma + ln_beta )
    int ln;
    const float *ln_gamma, *ln_beta;
    float ln_eps;
    int ln_post_relu;
    // dense 2D 'same' 3x3 convolution over [maps][img_h][img_w] pixel rows (K == 9): lets narrow layers run
    // on conv2d_tile_kernel, which needs no kernel map
    int img_h, img_w, img_maps;
    // the caller sizes bn_partial with eprecon_conv_desc_partial_rows (descriptor entry point), so kernels whose
    // workgroups do not cover 128 rows may be chosen
    int flex_partial;
    // dense-grid form of the 3x3x3 stride-1 convolution (conv3d_tile_kernel): vox_rank int32[gx][gy][gz] (z fastest) holds
    // the row of the voxel in a grid cell or -1; wq = the weights in MFMA operand order (pack_weights_kernel)
    const int32_t *vox_rank;
    int gx, gy, gz;
    const float *wq;    // ... in 32x32x2 operand order (pack_weights_kernel): B operands of the split-K and cross-workgroup kernels
    const float *wq16;  // ... in 16x16x4 operand order (pack_weights16_kernel): 16-row tile kernel, direct gather kernel
    int debug;  // always 0 in the library; timing probes build with it set: 1 no MFMA loop, 2 tile16: weights from the first offset, 4 no halo row loads
    int splitk_pipe;  // split-K kernel: 1 software-pipelined stages, 2 also B operands straight from the packed weights (wq)
    void *ws;         // caller's scratch (eprecon_conv_desc.workspace): partial sums of the cross-workgroup split-K kernel
    size_t ws_bytes;
    // BatchNorm form (c) (round 6): the producer adds its workgroups' per-channel sums to ORDER-INDEPENDENT
    // integer accumulators instead of (or beside) the per-workgroup summaries, and the consumer turns them into (scale, shift)
    // in its prologue — no finalize launch between two convolutions.  A block: int64[kBnCopies][ld][kBnWords] sums followed by
    // float[2][ld] (gamma, beta: written by the producer's first workgroup); *_c0 = first channel of this layer in the block.
    long long *bn_acc;
    int bn_acc_ld, bn_acc_c0;
    const float *bn_gamma, *bn_beta;      // the producer's BatchNorm parameters (copied into the block) or nullptr (1, 0)
    const long long *in_acc;              // consumer: the block its input's pending BatchNorm lives in
    int in_acc_ld, in_acc_c0;
    float in_eps;
};

// ---- BatchNorm form (c): exact, order-independent sums ------------------------------------------------------------------
// Per (copy, channel) seven int64 words: S1 = sum n_w mean_w and S2 = sum (M2_w + n_w mean_w^2) over the producer's workgroups
// w, each as THREE limbs of a 2^-40 fixed-point value (fraction, integer bits 0..39, integer bits 40.. signed), and the
// count.  Every add is a fire-and-forget 64-bit integer atomic (integer addition commutes: the result does not depend on the
// order the workgroups finish in, so inference stays run-to-run bit-identical); a limb sum cannot overflow (2^40 x 2^23 adds).
// The limbs are exact images of the doubles they come from, so sum n mean^2 - (sum n mean)^2 / N cancels exactly as the
// between-workgroup term of a Chan merge does.  kBnCopies copies spread the same-address atomics (21 ns each on this part).
constexpr int kBnCopies = 8, kBnWords = 7;
#ifndef EP_BN_ABL       // timing ablations of probe builds only: 1 no atomics, 2 no finish in the consumers (wrong results)
#define EP_BN_ABL 0
#endif

__device__ __forceinline__ void bn_limbs(double x, long long &l0, long long &l1, long long &l2)
{
    const double v = x * 1099511627776.0;                    // 2^40 (exact scaling)
    const double hi = floor(v * (1.0 / 1099511627776.0));    // floor(x)
    l0 = (long long)(v - hi * 1099511627776.0);              // fraction bits, in [0, 2^40) (below 2^-40: truncated)
    const long long h = (long long)hi;
    l1 = h & ((1ll << 40) - 1);
    l2 = h >> 40;
}

__device__ __forceinline__ void bn_acc_add(long long *acc, int ld, int c, int copy, float n, float mean, float m2)
{
    if (n <= 0.0f || (EP_BN_ABL & 1)) return;
    unsigned long long *a = reinterpret_cast<unsigned long long *>(acc) + ((size_t)copy * ld + c) * kBnWords;
    const double dn = (double)n, dm = (double)mean;
    long long l0, l1, l2;
    bn_limbs(dn * dm, l0, l1, l2);
    atomicAdd(a + 0, (unsigned long long)l0);
    if (EP_BN_ABL & 4) return;      // (one atomic per channel instead of seven)
    atomicAdd(a + 1, (unsigned long long)l1); atomicAdd(a + 2, (unsigned long long)l2);
    bn_limbs((double)m2 + dn * dm * dm, l0, l1, l2);
    atomicAdd(a + 3, (unsigned long long)l0); atomicAdd(a + 4, (unsigned long long)l1); atomicAdd(a + 5, (unsigned long long)l2);
    atomicAdd(a + 6, (unsigned long long)(long long)n);
}

// the BatchNorm of channel c of the block in affine form: y = x * scale + shift.  (The copies are walked ONE at a time: with the
// loop unrolled, 56 int64 loads in flight pushed the 1,024-thread split-K kernel, whose prologue inlines this, into scratch.)
__device__ __forceinline__ void bn_acc_affine(const long long *acc, int ld, int c, float eps, float &scale, float &shift)
{
    if (EP_BN_ABL & 2) { scale = 1.0f; shift = 0.0f; return; }
    long long w[kBnWords] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int k = 0; k < kBnCopies; ++k) {
        const long long *a = acc + ((size_t)k * ld + c) * kBnWords;
#pragma unroll
        for (int j = 0; j < kBnWords; ++j) w[j] += a[j];
    }
    const double two40 = 1099511627776.0;
    const double s1 = ((double)w[2] * two40 + (double)w[1]) + (double)w[0] * (1.0 / two40);
    const double s2 = ((double)w[5] * two40 + (double)w[4]) + (double)w[3] * (1.0 / two40);
    const double n = (double)w[6];
    const float *gb = reinterpret_cast<const float *>(acc + (size_t)kBnCopies * ld * kBnWords);
    const double mean = n > 0.0 ? s1 / n : 0.0;
    double var = n > 0.0 ? s2 / n - mean * mean : 0.0;     // biased variance
    if (var < 0.0) var = 0.0;
    const float sc = gb[c] / sqrtf((float)var + eps);
    scale = sc;
    shift = gb[ld + c] - (float)mean * sc;
}

// prologue of the gather kernels: (scale, shift) of the input channels -> sAff[0 .. cpad) / sAff[cpad .. 2 cpad), from the
// producer-finished vectors or from the accumulator block (zero beyond Cin)
template <int THREADS, bool ACC = true>
__device__ __forceinline__ void stage_in_affine(const ConvParams &p, float *sAff, int cpad, int tid)
{
    if (!p.in_scale) return;
    for (int c = tid; c < cpad; c += THREADS) {
        float sc = 0.0f, sh = 0.0f;
        if (c < p.Cin) {
            if (ACC && p.in_acc) bn_acc_affine(p.in_acc, p.in_acc_ld, p.in_acc_c0 + c, p.in_eps, sc, sh);
            else { sc = p.in_scale[c]; sh = p.in_shift[c]; }
        }
        sAff[c] = sc;
        sAff[cpad + c] = sh;
    }
}

// producer side: one thread per channel hands over its workgroup's summary; `first` (the launch's first workgroup) also leaves
// the BatchNorm parameters in the block
__device__ __forceinline__ void bn_acc_publish(const ConvParams &p, int col, int copy, bool first, float n, float mean, float m2)
{
    bn_acc_add(p.bn_acc, p.bn_acc_ld, p.bn_acc_c0 + col, copy & (kBnCopies - 1), n, mean, m2);
    if (first) {
        float *gb = reinterpret_cast<float *>(p.bn_acc + (size_t)kBnCopies * p.bn_acc_ld * kBnWords);
        gb[p.bn_acc_c0 + col] = p.bn_gamma ? p.bn_gamma[col] : 1.0f;
        gb[p.bn_acc_ld + p.bn_acc_c0 + col] = p.bn_beta ? p.bn_beta[col] : 0.0f;
    }
}

constexpr int kWaves = 4;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Chan et al. merge of two (count, mean, M2) summaries; the caller fixes the order
__device__ __forceinline__ void chan_merge(float &n_a, float &mean_a, float &m2_a, float n_b, float mean_b, float m2_b)
{
    if (n_b == 0.0f) return;
    if (n_a == 0.0f) {
        n_a = n_b; mean_a = mean_b; m2_a = m2_b;
        return;
    }
    const float n = n_a + n_b;
    const float d = mean_b - mean_a;
    mean_a = mean_a + d * (n_b / n);
    m2_a = m2_a + m2_b + d * d * (n_a * n_b / n);
    n_a = n;
}

// sparse_conv_direct.hip: the long-list 3x3x3 kernel on 16x16x4 MFMAs with operands straight from L2
bool direct16_ok(const ConvParams &p);
constexpr int kDirectRows = 128;          // output rows (and rows of a BatchNorm summary block) per workgroup
int launch_direct16(const ConvParams &p, hipStream_t st);
int direct16_partial_block_rows(const ConvParams &p);   // 128, or 32 when the persistent form takes the launch

}  // namespace epconv
