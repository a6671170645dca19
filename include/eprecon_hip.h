/*
 * eprecon_hip.h — C ABI of libeprecon_hip.so, the MI355X (gfx950) implementation of EPRecon's
 * per-fragment 3D hot path.
 *
 * The reference (zhen6618/EPRecon, paths below are relative to its root) has no FFI of its own:
 * its seam is the Python module API, and the arithmetic of the sparse layers lives in the
 * torchsparse / spconv CUDA extensions.  Each entry point here replaces one of those Python-level
 * operators (cited per function); the modules under eprecon_amd/ keep the reference's names and argument
 * meaning on top of this ABI, INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the library borrows it for
 *     the duration of the call (stream-ordered) and never frees or retains it, except for the
 *     opaque handles created by *_create and released by *_destroy;
 *   - `stream` is a hipStream_t passed as void*; NULL is the legacy default stream.  All work is
 *     enqueued on it; only functions documented as blocking synchronise it;
 *   - one caller thread per device (same as the reference: single Python thread, main.py:72);
 *   - return value: 0 = ok, 1 = "nothing to do" (the reference returns None and the caller bails
 *     out with a zero loss: ops/back_project.py:42-43, models/occupancy_initialization.py:107-108,
 *     :233-234), < 0 = error (EPRECON_ERR_* or -(1000 + hipError_t)).
 *   - coords are int32 rows (batch, x, y, z) in finest-voxel units, grouped by ascending batch
 *     index, exactly as the reference builds them (models/neucon_network.py:246-251).
 */
#ifndef EPRECON_HIP_H
#define EPRECON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPRECON_OK 0
#define EPRECON_EMPTY 1
#define EPRECON_ERR_ARG (-1)
#define EPRECON_ERR_WORKSPACE (-2)
#define EPRECON_ERR_UNSUPPORTED (-3)
#define EPRECON_ERR_HIP_BASE (-1000)

#define EPRECON_ABI_VERSION 1
int eprecon_abi_version(void);
/* name of the gfx target the kernels were compiled for, e.g. "gfx950" */
const char *eprecon_build_arch(void);

/* ------------------------------------------------------------------------------------------
 * Multi-view back-projection  (K1 project+mask+count, K2 stable compaction, K3 bilinear
 * gather-mean, K3' depth channel, K4 view variance)
 *
 * Replaces  Back_Project.forward            models/occupancy_initialization.py:189-261
 *           back_project                    ops/back_project.py:5-80          (mode MEAN_DEPTH)
 *           the sampling + mean/variance block of Occupancy_Initialization.forward
 *                                           models/occupancy_initialization.py:79-128 (VARIANCE)
 * ------------------------------------------------------------------------------------------ */
#define EPRECON_BP_MEAN 0       /* out_feats[n_valid, C]   = mean over visible views            */
#define EPRECON_BP_MEAN_DEPTH 1 /* out_feats[n_valid, C+1] = mean, normalised mean depth        */
#define EPRECON_BP_VARIANCE 2   /* out_feats[n_valid, C]   = population variance over views,
                                   out_mean[n_valid, C] (optional) = the view mean             */

#define EPRECON_LAYOUT_NCHW 0 /* feats f32[V, B, C, H, W]  (what the reference's backbone emits) */
#define EPRECON_LAYOUT_NHWC 1 /* feats f32[V, B, H, W, C]  (channels-last: skips the re-layout)   */

/* bytes of scratch the call needs (device memory, 256-byte aligned) */
size_t eprecon_back_project_workspace_bytes(int64_t n, int batch, int n_views, int channels,
                                            int height, int width, int feats_layout);

/*
 * Stream-ordered; does not synchronise.
 *   coords      int32[n,4]            origin   f32[batch,3]       krcam f32[V,batch,4,4]
 *   out_feats   f32[n, C or C+1]      (first n_valid rows are written, input order preserved)
 *   out_mean    f32[n, C] or NULL     out_coords int32[n,4]       count f32[n] (visible views)
 *   out_grid    f32[V, n_valid, 2] or NULL   normalised image coords of the valid voxels
 *   out_mask    u8 [V, n_valid]    or NULL   per-view visibility of the valid voxels
 *               (both packed with row stride n_valid, like the reference's im_grid / mask)
 *   n_valid_dev int32[1 + batch]      [0] = n_valid, [1 + b] = valid voxels of batch b
 */
int eprecon_back_project_async(const int32_t *coords, int64_t n, const float *origin, int batch,
                               float voxel_size, const float *feats, int feats_layout,
                               const float *krcam, int n_views, int channels, int height,
                               int width, int min_view, int mode, float *out_feats,
                               float *out_mean, int32_t *out_coords, float *count,
                               float *out_grid, uint8_t *out_mask, int32_t *n_valid_dev,
                               void *workspace, size_t workspace_bytes, void *stream);

/*
 * Blocking convenience: the call above, then copies n_valid_dev to n_valid_host[1 + batch] and
 * synchronises the stream (the reference synchronises at the same point: `torch.sum(valid_voxel)`
 * models/occupancy_initialization.py:231-233).  Returns EPRECON_EMPTY when some batch element
 * has fewer than `min_valid_per_batch` valid voxels (1 for Back_Project / back_project,
 * 1000 for the occupancy initialiser).
 */
int eprecon_back_project(const int32_t *coords, int64_t n, const float *origin, int batch,
                         float voxel_size, const float *feats, int feats_layout,
                         const float *krcam, int n_views, int channels, int height, int width,
                         int min_view, int mode, int min_valid_per_batch, float *out_feats,
                         float *out_mean, int32_t *out_coords, float *count, float *out_grid,
                         uint8_t *out_mask, int32_t *n_valid_dev, int32_t *n_valid_host,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * Per-kernel timing hook (what bench.py's roofline line is measured with).  While enabled, every
 * eprecon_back_project*_ call brackets its gather kernel with two hipEvents recorded on the
 * caller's stream.  eprecon_profile_gather_ms() synchronises on the last stop event and returns
 * the elapsed milliseconds of that kernel alone (blocking; < 0 when nothing was recorded).
 */
int eprecon_profile_enable(int on);
float eprecon_profile_gather_ms(void);

/* NCHW -> NHWC re-layout of a stack of feature maps: in f32[maps, C, H*W] -> out f32[maps, H*W, C] */
int eprecon_nchw_to_nhwc_async(const float *in, float *out, int maps, int channels, int hw,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EPRECON_HIP_H */
