// TSDF integration of depth frames into a voxel volume on gfx950 (the data-preparation side).
//
// Replaces the reference's per-sample CPU integration  TSDFVolumeTorch.integrate / integrate()
//                                                     tools/tsdf_fusion/fusion.py:440-485,551-575
// (called three times per sample by the data pipeline, datasets/transforms.py:286-297,375-387) and follows the
// semantics of its one in-tree CUDA kernel, the PyCUDA string `integrate`, tools/tsdf_fusion/fusion.py:67-142.
// The two differ in how a voxel centre reaches its pixel, so both are provided (variant argument):
//   0 "torch": cam = W2C @ [X,1] (k-ordered fp32 fma chain = torch's CPU matmul), px = rint(cam.x * fx / cam.z + cx),
//              valid z > 0, depth > 0                                   <- pinned bit-exact (tests/golden/tsdf_fusion.npz)
//   1 "cuda" : cam = R^T (X - t), px = roundf(fx * (cam.x / cam.z) + cx), valid z >= 0, depth != 0
// Shared update:  diff = depth - z;  skip if diff < -trunc;  dist = min(1, diff / trunc);
//                 w' = w + obs;  tsdf' = (tsdf * w + obs * dist) / w'.
// One thread per voxel (z fastest = coalesced over the [X,Y,Z] volume); ALL views of a fragment are integrated in
// one launch with the voxel's (tsdf, weight) held in registers, so a 9-view fragment costs one read and one
// write of the two volumes (16 B per voxel) instead of nine — HBM-bound by the contract, latency-bound at 96^3.
// The colour branch of the reference kernel is dead code (`return` at :129) and is not reproduced.
#include <math.h>

#include "common.hpp"

namespace {
using namespace ep;

constexpr int kMaxViews = 16;

struct TsdfViews {
    float cam[kMaxViews][12];  // rows 0..2 of world->camera (variant 0) / the camera pose (variant 1)
    float fx[kMaxViews], fy[kMaxViews], cx[kMaxViews], cy[kMaxViews];
    int n;
};

template <int VARIANT>
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(float *tsdf, float *weight, uint8_t *occ, int dx, int dy, int dz,
                                                             float ox, float oy, float oz, float vs, const float *depth,
                                                             int H, int W, TsdfViews v, float trunc, float obs)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= dx * dy * dz) return;
    const int iz = idx % dz, iy = (idx / dz) % dy, ix = idx / (dz * dy);
    const float X = __fadd_rn(__fmul_rn(vs, (float)ix), ox), Y = __fadd_rn(__fmul_rn(vs, (float)iy), oy),
                Z = __fadd_rn(__fmul_rn(vs, (float)iz), oz);
    float t = tsdf[idx], w = weight[idx];
    for (int k = 0; k < v.n; ++k) {
        const float *m = v.cam[k];
        float cxp, cyp, czp, px, py;
        bool front;
        if (VARIANT == 0) {
            cxp = __fmaf_rn(1.0f, m[3], __fmaf_rn(Z, m[2], __fmaf_rn(Y, m[1], __fmul_rn(m[0], X))));
            cyp = __fmaf_rn(1.0f, m[7], __fmaf_rn(Z, m[6], __fmaf_rn(Y, m[5], __fmul_rn(m[4], X))));
            czp = __fmaf_rn(1.0f, m[11], __fmaf_rn(Z, m[10], __fmaf_rn(Y, m[9], __fmul_rn(m[8], X))));
            px = rintf(__fadd_rn(__fdiv_rn(__fmul_rn(cxp, v.fx[k]), czp), v.cx[k]));
            py = rintf(__fadd_rn(__fdiv_rn(__fmul_rn(cyp, v.fy[k]), czp), v.cy[k]));
            front = czp > 0.0f;
        } else {
            // m = pose rows: pose[r][c] = m[4 r + c]; cam_c = pose[0][c] tx + pose[1][c] ty + pose[2][c] tz
            const float tx = __fsub_rn(X, m[3]), ty = __fsub_rn(Y, m[7]), tz = __fsub_rn(Z, m[11]);
            cxp = __fadd_rn(__fadd_rn(__fmul_rn(m[0], tx), __fmul_rn(m[4], ty)), __fmul_rn(m[8], tz));
            cyp = __fadd_rn(__fadd_rn(__fmul_rn(m[1], tx), __fmul_rn(m[5], ty)), __fmul_rn(m[9], tz));
            czp = __fadd_rn(__fadd_rn(__fmul_rn(m[2], tx), __fmul_rn(m[6], ty)), __fmul_rn(m[10], tz));
            px = roundf(__fadd_rn(__fmul_rn(v.fx[k], __fdiv_rn(cxp, czp)), v.cx[k]));
            py = roundf(__fadd_rn(__fmul_rn(v.fy[k], __fdiv_rn(cyp, czp)), v.cy[k]));
            front = !(czp < 0.0f);
        }
        if (!front || !(px >= 0.0f) || !(px < (float)W) || !(py >= 0.0f) || !(py < (float)H)) continue;  // NaN fails too
        const float d = depth[((size_t)k * H + (int)py) * W + (int)px];
        if (VARIANT == 0 ? !(d > 0.0f) : (d == 0.0f)) continue;
        const float diff = __fsub_rn(d, czp);
        if (diff < -trunc) continue;
        const float dist = fminf(1.0f, __fdiv_rn(diff, trunc));
        const float wn = __fadd_rn(w, obs);
        t = __fdiv_rn(__fadd_rn(__fmul_rn(w, t), __fmul_rn(obs, dist)), wn);
        w = wn;
    }
    tsdf[idx] = t;
    weight[idx] = w;
    if (occ) occ[idx] = (t < 0.999f && t > -0.999f && w > 1.0f) ? 1 : 0;   // datasets/transforms.py:295-297
}

}  // namespace

extern "C" int eprecon_tsdf_integrate_async(float *tsdf, float *weight, const int32_t *dims_host, const float *origin_host,
                                            float voxel_size, const float *depth, int n_views, int height, int width,
                                            const float *intr_host, const float *cam_host, float trunc, float obs_weight,
                                            int variant, uint8_t *occ_out, void *stream)
{
    if (!tsdf || !weight || !dims_host || !origin_host || !depth || !intr_host || !cam_host || n_views <= 0 ||
        height <= 0 || width <= 0 || !(trunc > 0.0f) || (variant != 0 && variant != 1))
        return EPRECON_ERR_ARG;
    const int64_t cells = (int64_t)dims_host[0] * dims_host[1] * dims_host[2];
    if (dims_host[0] <= 0 || dims_host[1] <= 0 || dims_host[2] <= 0 || cells > 0x7fffffff) return EPRECON_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    for (int v0 = 0; v0 < n_views; v0 += kMaxViews) {
        TsdfViews v;
        v.n = n_views - v0 < kMaxViews ? n_views - v0 : kMaxViews;
        for (int k = 0; k < v.n; ++k) {
            for (int e = 0; e < 12; ++e) v.cam[k][e] = cam_host[(size_t)(v0 + k) * 16 + e];
            const float *K = intr_host + (size_t)(v0 + k) * 9;
            v.fx[k] = K[0]; v.fy[k] = K[4]; v.cx[k] = K[2]; v.cy[k] = K[5];
        }
        const bool last = v0 + kMaxViews >= n_views;
        const float *dep = depth + (size_t)v0 * height * width;
        const dim3 grid((unsigned)ceil_div(cells, 256)), blk(256);
        if (variant == 0)
            hipLaunchKernelGGL(tsdf_integrate_kernel<0>, grid, blk, 0, st, tsdf, weight, last ? occ_out : nullptr, dims_host[0],
                               dims_host[1], dims_host[2], origin_host[0], origin_host[1], origin_host[2], voxel_size, dep,
                               height, width, v, trunc, obs_weight);
        else
            hipLaunchKernelGGL(tsdf_integrate_kernel<1>, grid, blk, 0, st, tsdf, weight, last ? occ_out : nullptr, dims_host[0],
                               dims_host[1], dims_host[2], origin_host[0], origin_host[1], origin_host[2], voxel_size, dep,
                               height, width, v, trunc, obs_weight);
        EP_LAUNCH_CHECK();
    }
    return EPRECON_OK;
}
