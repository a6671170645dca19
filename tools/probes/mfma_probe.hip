// What limits v_mfma_f32_32x32x2_f32 issue on gfx950?  Loops of N MFMAs per wave under different structures:
//   dep1      one accumulator (every MFMA depends on the previous one), operands in registers
//   dep2/4    2 / 4 independent accumulators
//   lds1      one accumulator, B operand from a conflict-free ds_read_b32 per MFMA (the convolution's inner loop)
//   lds2      two accumulators, B operand from LDS
// launched with 1, 2, 4 workgroups of 256 threads per CU resident (occupancy via dynamic LDS size).
// Prints TFLOP/s against the 157.3 TF fp32 MFMA peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void probe(float *out, int iters)
{
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) smem[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float a = 1.0f + lane, b = 0.5f * lane;
    const float *w = smem + (lane & 31) + 4 * (lane >> 5) * 32;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                float bb = LDS ? w[((s * NACC + t) & 31) * 32] : b;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[t], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char *name, int wg_per_cu, float *out)
{
    const int iters = 2000;
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : wg_per_cu == 2 ? 64 * 1024 : wg_per_cu == 4 ? 36 * 1024 : 16 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<NACC, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, LDS>), dim3(grid), dim3(256), lds, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, LDS>), dim3(grid), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 16 * NACC * 4096.0;
    printf("%-6s wg/CU %d: %7.3f ms  %6.1f TF (%.0f %% of 157.3)\n", name, wg_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int occ : {1, 2, 4, 8}) {
        run<1, false>("dep1", occ, out);
        run<2, false>("dep2", occ, out);
        run<4, false>("dep4", occ, out);
        run<1, true>("lds1", occ, out);
        run<2, true>("lds2", occ, out);
    }
    return 0;
}
