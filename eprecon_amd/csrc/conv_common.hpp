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
};

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
