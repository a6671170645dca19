/*
 * ORACLE — test infrastructure, not product code.
 *
 * Plain-C restatement of the reference's multi-view back-projection.  Only tests/, the smoke
 * check in __graft_entry__.py and the cpu_baseline leg of bench.py may load this library; the
 * shipped path (eprecon_amd/) never does.
 *
 * Follows (paths under /root/reference):
 *   models/occupancy_initialization.py:205-259   Back_Project.forward   (mode MEAN)
 *   ops/back_project.py:13-78                    back_project           (mode MEAN_DEPTH)
 *   models/occupancy_initialization.py:79-128    view mean / variance   (mode VARIANCE)
 *
 * Arithmetic contract, all fp32:
 *   X   = float(c) * voxel_size + origin_b                      (mul, then add; no fusion)
 *   p_r = fma(P[r][3], 1, fma(P[r][2], Z, fma(P[r][1], Y, P[r][0] * X)))   k-ordered fma chain.
 *         This is bit-identical to torch's CPU bmm for the [4x4]@[4xN] product the reference
 *         issues (checked in the build container on 23.9 M elements, 0 mismatches; a non-fused
 *         sum differs on 25 % of them) and to the fp32 MFMA / v_fmac chain on gfx950.
 *   u = p_x / p_z, v = p_y / p_z                                 (IEEE division)
 *   gx = (2*u)/(W-1) - 1, gy = (2*v)/(H-1) - 1
 *   visible_v = |gx| <= 1 && |gy| <= 1 && p_z > 0 ;  count = sum_v visible_v   (stored as float)
 *   valid = count >= min_view ; output rows = input order filtered by valid
 *   sample: ix = ((gx+1)/2)*(W-1), bilinear with zero padding, align_corners=True semantics;
 *           views that are not visible contribute 0; divide by max(count,1).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  -ffp-contract=off matters:
 * every fma in this file is written explicitly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

enum { MODE_MEAN = 0, MODE_MEAN_DEPTH = 1, MODE_VARIANCE = 2 };

typedef struct {
    float gx, gy, pz;
    int visible;
} proj_t;

static inline proj_t project_one(const float *P, float X, float Y, float Z, int H, int W)
{
    proj_t r;
    float px = fmaf(P[3], 1.0f, fmaf(P[2], Z, fmaf(P[1], Y, P[0] * X)));
    float py = fmaf(P[7], 1.0f, fmaf(P[6], Z, fmaf(P[5], Y, P[4] * X)));
    float pz = fmaf(P[11], 1.0f, fmaf(P[10], Z, fmaf(P[9], Y, P[8] * X)));
    float u = px / pz;
    float v = py / pz;
    r.gx = (2.0f * u) / (float)(W - 1) - 1.0f;
    r.gy = (2.0f * v) / (float)(H - 1) - 1.0f;
    r.pz = pz;
    r.visible = (fabsf(r.gx) <= 1.0f) && (fabsf(r.gy) <= 1.0f) && (pz > 0.0f);
    return r;
}

/* bilinear sample of one view's NCHW map at normalised (gx, gy); adds w * value into acc[C]
 * (acc may be NULL, then writes into out[C]) */
static inline void bilinear_chw(const float *map, int C, int H, int W, float gx, float gy, float *out)
{
    float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    float x0f = floorf(ix), y0f = floorf(iy);
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    float wx1 = ix - x0f, wx0 = (x0f + 1.0f) - ix;
    float wy1 = iy - y0f, wy0 = (y0f + 1.0f) - iy;
    float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
    int in00 = x0 >= 0 && x0 < W && y0 >= 0 && y0 < H;
    int in10 = x1 >= 0 && x1 < W && y0 >= 0 && y0 < H;
    int in01 = x0 >= 0 && x0 < W && y1 >= 0 && y1 < H;
    int in11 = x1 >= 0 && x1 < W && y1 >= 0 && y1 < H;
    size_t plane = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        const float *m = map + (size_t)c * plane;
        float s = 0.0f;
        if (in00) s += m[(size_t)y0 * W + x0] * w00;
        if (in10) s += m[(size_t)y0 * W + x1] * w10;
        if (in01) s += m[(size_t)y1 * W + x0] * w01;
        if (in11) s += m[(size_t)y1 * W + x1] * w11;
        out[c] = s;
    }
}

/*
 * coords    int32[N,4] (b,x,y,z) finest-voxel units, grouped by ascending batch index
 * origin    f32[B,3]
 * feats     f32[V,B,C,H,W]
 * krcam     f32[V,B,4,4]
 * out_feats f32[>=N, Cout]  Cout = C (MEAN, VARIANCE) or C+1 (MEAN_DEPTH)
 * out_mean  f32[>=N, C] or NULL (VARIANCE only: the per-voxel view mean)
 * out_coords int32[>=N,4]
 * count     f32[N]
 * out_grid  f32[V, n_valid, 2] or NULL (written with row stride n_valid, so pass 2 only)
 * out_mask  u8 [V, n_valid]    or NULL
 * returns n_valid (>= 0), or -1 when some batch has no valid voxel (reference returns None)
 */
int64_t eprecon_oracle_back_project(const int32_t *coords, int64_t n, const float *origin, int B,
                                    float voxel_size, const float *feats, const float *krcam, int V,
                                    int C, int H, int W, int min_view, int mode, float *out_feats,
                                    float *out_mean, int32_t *out_coords, float *count,
                                    float *out_grid, uint8_t *out_mask)
{
    if (V > 32) return -2;
    int64_t *slot = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *per_batch = (int64_t *)calloc((size_t)B, sizeof(int64_t));

#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int b = coords[4 * i];
        float X = (float)coords[4 * i + 1] * voxel_size + origin[3 * b + 0];
        float Y = (float)coords[4 * i + 2] * voxel_size + origin[3 * b + 1];
        float Z = (float)coords[4 * i + 3] * voxel_size + origin[3 * b + 2];
        int cnt = 0;
        for (int v = 0; v < V; ++v) {
            proj_t p = project_one(krcam + ((size_t)v * B + b) * 16, X, Y, Z, H, W);
            cnt += p.visible;
        }
        count[i] = (float)cnt;
        slot[i] = (count[i] >= (float)min_view) ? 1 : 0;
    }
    int64_t n_valid = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t is_valid = slot[i];
        slot[i] = is_valid ? n_valid : -1;
        n_valid += is_valid;
        if (is_valid) per_batch[coords[4 * i]] += 1;
    }
    int empty_batch = 0;
    for (int b = 0; b < B; ++b) empty_batch |= (per_batch[b] == 0);
    free(per_batch);
    if (empty_batch) {
        free(slot);
        return -1;
    }

    int Cout = (mode == MODE_MEAN_DEPTH) ? C + 1 : C;
    size_t map_sz = (size_t)C * H * W;
    float *depth = NULL;
    if (mode == MODE_MEAN_DEPTH) depth = (float *)malloc(sizeof(float) * (size_t)n_valid);

#pragma omp parallel
    {
        float *samp = (float *)malloc(sizeof(float) * (size_t)C * (size_t)V);
        float *acc = (float *)malloc(sizeof(float) * (size_t)C);
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            int64_t o = slot[i];
            if (o < 0) continue;
            int b = coords[4 * i];
            float X = (float)coords[4 * i + 1] * voxel_size + origin[3 * b + 0];
            float Y = (float)coords[4 * i + 2] * voxel_size + origin[3 * b + 1];
            float Z = (float)coords[4 * i + 3] * voxel_size + origin[3 * b + 2];
            memcpy(out_coords + 4 * o, coords + 4 * i, 4 * sizeof(int32_t));
            for (int c = 0; c < C; ++c) acc[c] = 0.0f;
            float zsum = 0.0f;
            int cnt = 0;
            uint32_t vis = 0;
            for (int v = 0; v < V; ++v) {
                proj_t p = project_one(krcam + ((size_t)v * B + b) * 16, X, Y, Z, H, W);
                if (out_grid) {
                    out_grid[((size_t)v * n_valid + o) * 2 + 0] = p.gx;
                    out_grid[((size_t)v * n_valid + o) * 2 + 1] = p.gy;
                }
                if (out_mask) out_mask[(size_t)v * n_valid + o] = (uint8_t)p.visible;
                float *s = samp + (size_t)v * C;
                if (p.visible) {
                    bilinear_chw(feats + ((size_t)v * B + b) * map_sz, C, H, W, p.gx, p.gy, s);
                    for (int c = 0; c < C; ++c) acc[c] += s[c];
                    zsum += p.pz;
                    cnt += 1;
                    vis |= 1u << v;
                } else {
                    for (int c = 0; c < C; ++c) s[c] = 0.0f;
                }
            }
            float denom = (float)(cnt > 0 ? cnt : 1);
            float *of = out_feats + (size_t)o * Cout;
            if (mode == MODE_VARIANCE) {
                /* models/occupancy_initialization.py:124-128: population variance over the
                 * visible views; the divisor is the raw visible count (>= min_view >= 1 here) */
                float nvis = (float)cnt;
                for (int c = 0; c < C; ++c) {
                    float mean = acc[c] / nvis;
                    float q = 0.0f;
                    for (int v = 0; v < V; ++v) {
                        if (vis & (1u << v)) {
                            float d = samp[(size_t)v * C + c] - mean;
                            q += d * d;
                        }
                    }
                    of[c] = q / nvis;
                    if (out_mean) out_mean[(size_t)o * C + c] = mean;
                }
            } else {
                for (int c = 0; c < C; ++c) of[c] = acc[c] / denom;
                if (mode == MODE_MEAN_DEPTH) depth[o] = zsum / denom;
            }
        }
        free(samp);
        free(acc);
    }

    if (mode == MODE_MEAN_DEPTH) {
        /* ops/back_project.py:69-75: mean over d > 0, L2 norm (not std) + 1e-5, zero where d <= 0.
         * The reference normalises per batch inside its batch loop. */
        int64_t start = 0;
        while (start < n_valid) {
            int b = out_coords[4 * start];
            int64_t end = start;
            while (end < n_valid && out_coords[4 * end] == b) ++end;
            double sum = 0.0;
            int64_t m = 0;
            for (int64_t o = start; o < end; ++o)
                if (depth[o] > 0.0f) { sum += depth[o]; ++m; }
            float mu = (float)(sum / (double)(m > 0 ? m : 1));
            double ss = 0.0;
            for (int64_t o = start; o < end; ++o)
                if (depth[o] > 0.0f) { double d = (double)(depth[o] - mu); ss += d * d; }
            float sigma = (float)sqrt(ss) + 1e-5f;
            for (int64_t o = start; o < end; ++o)
                out_feats[(size_t)o * Cout + C] = depth[o] > 0.0f ? (depth[o] - mu) / sigma : 0.0f;
            start = end;
        }
        free(depth);
    }
    free(slot);
    return n_valid;
}

void eprecon_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int eprecon_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* r = W2AC[:3,:] . [c * vs + origin, 1]  (models/neucon_network.py:387-398): the reference computes
 * a [N,4] @ [4,3] matmul; like the projection above it is restated as the k-ordered fma chain.
 * out f32[n,4] = (x, y, z, batch) — torchsparse PointTensor order. */
void eprecon_oracle_aligned_coords(const int32_t *coords, int64_t n, const float *origin, int B,
                                   float voxel_size, const float *w2ac, float *out)
{
    (void)B;
    for (int64_t i = 0; i < n; ++i) {
        int b = coords[4 * i];
        float X = (float)coords[4 * i + 1] * voxel_size + origin[3 * b + 0];
        float Y = (float)coords[4 * i + 2] * voxel_size + origin[3 * b + 1];
        float Z = (float)coords[4 * i + 3] * voxel_size + origin[3 * b + 2];
        const float *M = w2ac + 16 * b;
        for (int j = 0; j < 3; ++j)
            out[4 * i + j] = fmaf(1.0f, M[4 * j + 3], fmaf(Z, M[4 * j + 2], fmaf(Y, M[4 * j + 1], X * M[4 * j])));
        out[4 * i + 3] = (float)b;
    }
}
