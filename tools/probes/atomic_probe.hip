// probe: what do fire-and-forget 64-bit integer atomics to a handful of addresses cost at the tail of every workgroup?
// (the question behind "BatchNorm sums by integer atomics instead of summary rows + a finalize launch")
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void work_kernel(const float *x, float *y, unsigned long long *acc, int C, int mode, int n)
{
    const int row0 = blockIdx.x * 128;
    float s = 0.f;
    for (int r = threadIdx.x; r < 128 * 8; r += 256) {      // some memory work per workgroup (~4 KB)
        const int i = row0 * 8 + r;
        if (i < n) s += x[i];
    }
    if (row0 * 8 + (int)threadIdx.x < n) y[row0 * 8 + threadIdx.x] = s;
    if (mode == 1 && (int)threadIdx.x < 2 * C) atomicAdd(acc + threadIdx.x, (unsigned long long)(threadIdx.x + 1));
    if (mode == 2 && (int)threadIdx.x < 2 * C) atomicAdd(acc + (blockIdx.x % 8) * 2 * C + threadIdx.x, (unsigned long long)(threadIdx.x + 1));
}

int main()
{
    const int blocks = 1350, C = 64, n = blocks * 128 * 8;
    float *x, *y;
    unsigned long long *acc;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&acc, 8 * 2 * C * 8);
    hipMemset(x, 0, n * 4); hipMemset(acc, 0, 8 * 2 * C * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, 0, x, y, acc, C, mode, n);
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, 0, x, y, acc, C, mode, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.2f us per launch of %d workgroups\n", mode,
               mode == 0 ? "no atomics" : mode == 1 ? "128 atomics per workgroup to 128 addresses" : "the same, spread over 8 copies", ms / 50 * 1e3, blocks);
    }
    return 0;
}
