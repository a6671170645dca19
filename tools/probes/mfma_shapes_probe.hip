// Issue rate of the fp32 MFMA shapes the direct gather kernel mixes (round 6: the 16 + 8 column split runs the last 8 columns
// on v_mfma_f32_4x4x1_16B_f32): loops of independent MFMAs per wave, 1 / 2 / 4 waves per SIMD.
//   m16     four accumulators of v_mfma_f32_16x16x4_f32 in rotation                    2048 flop / instruction
//   m4      eight accumulators of v_mfma_f32_4x4x1_16B_f32 in rotation                  512 flop / instruction
//   mix     the kernel's pattern per gathered register pair: 2 x 16x16x4 + 4 x 4x4x1    (2 x 2048 + 4 x 512 flop)
// Prints TFLOP/s against the 157.3 TF fp32 MFMA peak and cycles per instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x4 a16[4], a4[8];
    for (int t = 0; t < 4; ++t) a16[t] = f32x4{0, 0, 0, 0};
    for (int t = 0; t < 8; ++t) a4[t] = f32x4{0, 0, 0, 0};
    float a = 1.0f + lane, b = 0.5f * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE == 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) a16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[t], 0, 0, 0);
            } else if (MODE == 1) {
#pragma unroll
                for (int t = 0; t < 8; ++t) a4[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a4[t], 0, 0, 0);
            } else {
                a16[(2 * s) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[(2 * s) & 3], 0, 0, 0);
                a16[(2 * s + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[(2 * s + 1) & 3], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[(4 * s + t) & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a4[(4 * s + t) & 7], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) s += a16[t][0] + a16[t][3];
    for (int t = 0; t < 8; ++t) s += a4[t][0] + a4[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int wg_per_cu, float *out)
{
    const int iters = 4000;
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : wg_per_cu == 2 ? 64 * 1024 : 36 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(256), lds, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_iter_flop = MODE == 0 ? 8 * 4 * 2048.0 : MODE == 1 ? 8 * 8 * 512.0 : 8 * (2 * 2048.0 + 4 * 512.0);
    const double per_iter_inst = MODE == 0 ? 32 : MODE == 1 ? 64 : 48;
    const double flops = (double)grid * 4 * iters * per_iter_flop;
    // cycles per instruction per SIMD: each SIMD runs wg_per_cu waves
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * per_iter_inst * wg_per_cu);
    printf("%-4s waves/SIMD %d: %7.3f ms  %6.1f TF (%3.0f %% of 157.3)  %5.1f cycles per instruction per SIMD at 2.4 GHz\n", name, wg_per_cu, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, cyc);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    for (int occ : {1, 2, 4}) {
        run<0>("m16", occ, out);
        run<1>("m4", occ, out);
        run<2>("mix", occ, out);
    }
    return 0;
}
