// probe: how many 64-byte vector-L1 segments ("lines" in DESIGN.md 3a / bench.py roofline.l1) can a CU of gfx950 serve per clock
// for the access pattern of bp_gather_mlp_kernel — 16-byte loads, six consecutive lanes covering one 96-byte texel (24 fp32
// channels), texels scattered — and for the patterns it could be traded for (128-byte padded texels, 64-byte texels, one
// contiguous 1 KB run per wave instruction)?  VERDICT r04 item 5: the "1 line / clk / CU" peak of the L1 roofline was an
// assumption; this calibrates it.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/l1_line_rate tools/probes/l1_line_rate.hip && tools/probes/l1_line_rate
//
// Every lane runs a chain-free stream of loads (8 in flight), texel indices from a per-lane LCG (cheap VALU next to the loads).
// Footprints: 12 KB (L1-resident: every CU reads the same table), 2 MB (L2-resident per XCD), 16.6 MB (= the nine 120x160x24
// maps of the timed level: what the gather actually walks).  Segments per wave instruction are exact by construction:
//   mode 0  96-byte texels, 60 active lanes = 10 texels, each texel spans exactly 2 segments            -> 20
//   mode 1  128-byte padded texels (8 lanes, 96 useful bytes: lanes 6, 7 idle), 8 texels x 2 segments   -> 16
//   mode 2  64-byte texels (4 lanes), 16 texels x 1 segment                                             -> 16
//   mode 3  one contiguous 1 KB run per instruction                                                     -> 16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe_kernel(const char *table, unsigned n_texels, int iters, float *sink)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    constexpr unsigned kLanes = MODE == 0 ? 6u : MODE == 1 ? 8u : MODE == 2 ? 4u : 64u;   // lanes that share a texel
    constexpr unsigned kPitch = MODE == 0 ? 96u : MODE == 1 ? 128u : MODE == 2 ? 64u : 1024u;
    const unsigned group = lane / kLanes, sub = lane % kLanes;
    const bool active = MODE == 0 ? lane < 60u : MODE == 1 ? sub < 6u : true;
    if (!active) return;                                                  // (idle lanes leave: no per-load branch in the loop)
    unsigned state = (wave * 64u + group) * 2654435761u + 12345u;       // one stream per (wave, texel group)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            state = state * 1664525u + 1013904223u;
            const unsigned texel = __umulhi(state, n_texels);      // uniform in [0, n_texels): one instruction, no division
            const char *p = table + (size_t)texel * kPitch + sub * 16u;
            v[u] = *reinterpret_cast<const f32x4 *>(p);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;      // never true: keeps the loads alive
}

template <int MODE>
static void run(const char *name, const char *table, size_t footprint, int segs_per_instr, int useful_bytes_per_instr, float *sink)
{
    constexpr unsigned pitch = MODE == 0 ? 96u : MODE == 1 ? 128u : MODE == 2 ? 64u : 1024u;
    const unsigned n_texels = (unsigned)(footprint / pitch);
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(probe_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, table, n_texels, iters, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, table, n_texels, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double s = ms * 1e-3 / reps;
    const double instr = (double)blocks * 4 * iters;                     // wave-level load instructions per launch
    const double segs = instr * segs_per_instr;
    printf("%-46s footprint %9.1f KB  %7.1f us  %6.1f G wave-loads/s  %7.1f G segments/s = %5.3f per clk per CU (256 CUs, 2.4 GHz)  "
           "%6.2f TB/s useful\n", name, footprint / 1024.0, s * 1e6, instr / s / 1e9, segs / s / 1e9, segs / s / (256 * 2.4e9),
           instr * useful_bytes_per_instr / s / 1e12);
}

// ---- FETCH_SIZE calibration (`l1_line_rate calib` under rocprofv3 --pmc FETCH_SIZE): two launches whose HBM-side fetch is known ----
// stream: every wave instruction reads its own contiguous 1 KB exactly once (1 GiB in total): fetch = 1 GiB;
// gather: the gather's access shape — six lanes x 16 bytes per 96-byte texel — over a 1.5 GiB table, every texel visited exactly
// once in a scrambled order (an odd multiplier is a bijection on 2^24 indices): no reuse inside an L1 or an XCD's 4 MB L2, so
// every 128-byte line a texel touches is fetched: texels at byte offsets 0, 96, 64, 32 (mod 128) span 1, 2, 2, 1 lines ->
// 1.5 lines = 192 bytes fetched per 96-byte texel when nothing is retained, 128 when a line's second toucher still finds it.
__global__ __launch_bounds__(256) void calib_stream_kernel(const char *table, float *sink)
{
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(table + wave * 1024 + (threadIdx.x & 63) * 16);
    if (v.x == 123.456f) sink[0] = v.y;
}
__global__ __launch_bounds__(240) void calib_gather_kernel(const char *table, unsigned log2_texels, float *sink)
{
    const unsigned g = blockIdx.x * 40u + threadIdx.x / 6u;             // one texel per 6-lane group, 40 groups per workgroup
    const unsigned texel = (g * 2654435761u) & ((1u << log2_texels) - 1u);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(table + (size_t)texel * 96 + (threadIdx.x % 6u) * 16);
    if (v.x == 123.456f) sink[0] = v.y;
}

static int calibrate()
{
    const unsigned log2_texels = 24;
    const size_t gather_bytes = ((size_t)1 << log2_texels) * 96, stream_bytes = (size_t)1 << 30;
    char *table;
    float *sink;
    hipMalloc(&table, gather_bytes);
    hipMalloc(&sink, 64);
    hipMemset(table, 0, gather_bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(calib_stream_kernel, dim3((unsigned)(stream_bytes / 1024 / 4)), dim3(256), 0, 0, table, sink);
    hipLaunchKernelGGL(calib_gather_kernel, dim3((1u << log2_texels) / 40u), dim3(240), 0, 0, table, log2_texels, sink);
    hipDeviceSynchronize();
    printf("# calibration launches: calib_stream_kernel reads %zu bytes once; calib_gather_kernel reads %u texels of 96 bytes once "
           "(%zu bytes useful, %zu bytes of distinct 128-byte lines, %zu if every texel re-fetches the lines it spans)\n",
           stream_bytes, ((1u << log2_texels) / 40u) * 40u, (size_t)((1u << log2_texels) / 40u) * 40u * 96, gather_bytes,
           (size_t)((1u << log2_texels) / 40u) * 40u * 192);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'c') return calibrate();
    const size_t big = 9ull * 120 * 160 * 24 * 4 + 4096;      // 16.6 MB: the maps of the timed level
    char *table;
    float *sink;
    hipMalloc(&table, 32ull << 20);
    hipMalloc(&sink, 64);
    hipMemset(table, 0, 32ull << 20);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("# tools/probes/l1_line_rate.hip on %s (%d CUs, %d MHz)\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const size_t fp[3] = {12288, 2u << 20, big};
    const char *fpn[3] = {"L1-resident", "L2-resident", "16.6 MB (the level's maps)"};
    for (int f = 0; f < 3; ++f) {
        printf("## %s\n", fpn[f]);
        run<0>("mode 0: 96-byte texels, 6 lanes each", table, fp[f], 20, 960, sink);
        run<1>("mode 1: 128-byte padded texels, 6 of 8 lanes", table, fp[f], 16, 768, sink);
        run<2>("mode 2: 64-byte texels, 4 lanes each", table, fp[f], 16, 1024, sink);
        run<3>("mode 3: contiguous 1 KB per wave instruction", table, fp[f], 16, 1024, sink);
    }
    return 0;
}
