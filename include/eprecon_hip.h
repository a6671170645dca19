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
 * on == 2 arms a one-shot: only the next call's gather kernel is bracketed, later calls leave the
 * recorded pair alone (how bench.py singles out the dense 96^3 level inside a queued step).
 */
int eprecon_profile_enable(int on);
float eprecon_profile_gather_ms(void);
/* family of the gather kernel the recorded pair brackets ("bp_gather_mlp_kernel" / "bp_gather_kernel") */
const char *eprecon_profile_gather_kernel(void);

/*
 * The same for the gather-GEMM convolution (bench.py's `roofline_conv`): arm a one-shot — the next
 * sparse-convolution launch with this (kvol, cin, cout) on a list of at least min_rows rows is
 * bracketed by two events on its stream.  eprecon_profile_conv_ms blocks on the stop event and returns
 * the milliseconds (< 0: nothing recorded), the list length and the name of the kernel family chosen.
 */
int eprecon_profile_conv_arm(int kvol, int cin, int cout, int64_t min_rows);
float eprecon_profile_conv_ms(int64_t *rows_out, const char **kernel_out);
/* live (output row, kernel offset) pairs of the bracketed launch (2 * pairs * cin * cout = its algorithmic flops),
 * counted on the launch stream behind the stop event; blocks; -1 when nothing was recorded */
int64_t eprecon_profile_conv_pairs(void);
/* the (row, offset) pairs that launch issued MFMAs for: the direct kernel skips an offset for the 32 rows of a wave when none
 * of them has a neighbour there, so this is 32 x the live (32-row group, offset) pairs; 0 for the kernels that walk every offset
 * of every row (executed = kvol * rows) */
int64_t eprecon_profile_conv_executed_pairs(void);
/* name of the kernel family the most recent convolution launch of this process went to ("spconv_direct16_kernel",
 * "spconv_splitk_kernel", "spconv_wide_kernel", "conv3d_tile16_kernel", ...): lets callers and tests assert the selection */
const char *eprecon_profile_last_conv_kernel(void);
/* stage marker for rocprofv3 kernel traces: an empty launch of (id + 1) workgroups on `stream` (tools/trace_cfg4_layers.py) */
int eprecon_profile_mark_async(int id, void *stream);

/* NCHW -> NHWC re-layout of a stack of feature maps: in f32[maps, C, H*W] -> out f32[maps, H*W, C] */
int eprecon_nchw_to_nhwc_async(const float *in, float *out, int maps, int channels, int hw,
                               void *stream);
/* The same for the per-view maps of up to three pyramid levels in ONE launch: src[l][v] f32[C_l, hw_l] (one contiguous map per
 * view, as models/occupancy_initialization.py:79-90 receives them: features_all[view][level][batch]) -> dst[l] f32[n_views * hw_l,
 * C_l] (pixel rows, views stacked).  Replaces torch.stack per level + the channels-last copy in front of the 2D fusion stack. */
typedef struct eprecon_views_desc {
    const float *src[3][16];
    float *dst[3];
    int32_t channels[3]; int32_t hw[3];
    int32_t levels; int32_t n_views;
} eprecon_views_desc;
int eprecon_views_to_rows_async(const eprecon_views_desc *desc, void *stream);

/* ------------------------------------------------------------------------------------------
 * Hash grid over voxel coordinates  (K6 / K7)
 *
 * Replaces  torchsparse F.sphash + F.sphashquery   ops/torchsparse_utils.py:19-21,44-50,73-79
 *           torch.unique(hash) voxel numbering     ops/torchsparse_utils.py:20-22
 * (torchsparse is an un-vendored dependency of the reference, README.md:19; semantics restated
 * in DESIGN.md.)  A table is caller-owned device memory of eprecon_hash_table_bytes(capacity)
 * bytes, capacity = eprecon_hash_capacity(n) (a power of two >= 2n).  Keys are exact 64-bit
 * packings of (b, x, y, z): |coordinate| < 2^19 - 1, 0 <= b <= 14.  `quantum` q >= 1 maps every
 * spatial coordinate to floor(c / q) * q before hashing (q = 2 * tensor_stride builds the strided
 * coordinate set of a k2s2 convolution).
 * ------------------------------------------------------------------------------------------ */
uint32_t eprecon_hash_capacity(int64_t n);
size_t eprecon_hash_table_bytes(uint32_t capacity);
/* value of a key = smallest input row that carries it */
int eprecon_hash_build_async(const int32_t *coords, int64_t n, int quantum, void *table,
                             uint32_t capacity, void *stream);
/* ... with the row count on the device: only the first min(n_cap, *n_dev) rows are inserted (the table is sized by n_cap) */
int eprecon_hash_build_dn_async(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, int quantum, void *table,
                                uint32_t capacity, void *stream);
/* out_index[i] = value stored for queries[i] (int32[m,4] bxyz), or -1 */
int eprecon_hash_query_async(const void *table, uint32_t capacity, const int32_t *queries, int64_t m,
                             int quantum, int32_t *out_index, void *stream);
/* blocking: EPRECON_OK, EPRECON_ERR_UNSUPPORTED (a key was out of range) or
 * EPRECON_ERR_WORKSPACE (table full) */
int eprecon_hash_status(const void *table, void *stream);

/* Unique (quantised) coordinates in FIRST-OCCURRENCE order:
 *   inverse[i]           id of row i's voxel                        int32[n]
 *   unique_coords[id]    its quantised (b,x,y,z)                     int32[n,4] (first n_unique rows)
 *   n_unique_dev         int32[1]
 * afterwards the table maps key -> voxel id (so eprecon_hash_query_async returns ids). */
size_t eprecon_unique_workspace_bytes(int64_t n);
/* exclusive prefix sum of int32 (the device scan behind every compaction / unique numbering of the path: torch.nonzero,
 * boolean-mask indexing and torch.unique at models/neucon_network.py:298-318,454-507, ops/torchsparse_utils.py:20-22):
 * out[i] = in[0] + ... + in[i - 1]; *total (optional, device) = the sum of all n; scratch int32[ceil(n / 2048)] */
int eprecon_exclusive_scan_async(const int32_t *in, int64_t n, int32_t *out, int32_t *total, int32_t *scratch, void *stream);
int eprecon_unique_coords_async(const int32_t *coords, int64_t n, int quantum, void *table,
                                uint32_t capacity, int32_t *inverse, int32_t *unique_coords,
                                int32_t *n_unique_dev, void *workspace, size_t workspace_bytes,
                                void *stream);

/* The same with the row count ON THE DEVICE ("outputs written into caller-allocated buffers sized by the static
 * caps, with the actual count returned through a device counter"): the live count is min(n_cap, *n_dev), buffers, the
 * table and the workspace are sized for n_cap, launches cover n_cap rows.  Lets a chain of data-dependent steps (quantise ->
 * unique -> unique of the coarser stride ...) be queued without a host round trip between them; the caller reads all
 * counts back once. */
int eprecon_unique_coords_dn_async(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, int quantum, void *table,
                                   uint32_t capacity, int32_t *inverse, int32_t *unique_coords, int32_t *n_unique_dev,
                                   void *workspace, size_t workspace_bytes, void *stream);
/*
 * The voxel sets of a point cloud at tensor strides 1, 2, 4 (levels <= 3) numbered back to back with every count on the device
 * (SPVCNN: models/modules.py:148-175 voxelises at stride 1, its two k2s2 stages build strides 2 and 4): level l numbers the
 * unique rows of level l - 1 (level 0: the first min(n_cap, *n_dev) rows of `coords`; n_dev NULL: all n_cap) at quantum 2^l into
 * tables[l] (capacities[l] slots, eprecon_hash_capacity(n_cap)), inverse[l] int32[n_cap], unique_coords[l] int32[n_cap,4]; the
 * count of level l is written next to its table's status word (int32 tables[l][1]).  The tables are reset by ONE launch.
 * summary (optional, device int32[2 levels]): receives (status, count) of every level side by side — what the host reads.
 */
int eprecon_unique_hierarchy_dn_async(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, int levels, void *const *tables,
                                      const uint32_t *capacities, int32_t *const *inverse, int32_t *const *unique_coords,
                                      int32_t *summary, void *workspace, size_t workspace_bytes, void *stream);

/*
 * eprecon_kernel_map_async(ksize 3) for a voxel set queried against ITS OWN table — coords int32[n,4] are pairwise distinct and
 * row i is the table's value for its key (what eprecon_unique_coords*_async / eprecon_unique_hierarchy_dn_async leave behind) —
 * with half the hash probes: the map of a submanifold 3x3x3 convolution (models/modules.py:19-23,181, spnn.Conv3d stride 1) is
 * symmetric (offset -o from j reaches i exactly when +o from i reaches j).  Same int32[27][n] table, bit for bit.
 */
int eprecon_kernel_map_self_async(const void *table, uint32_t capacity, const int32_t *coords, int64_t n, int stride, int32_t *nbr,
                                  void *stream);
/* ------------------------------------------------------------------------------------------
 * Kernel maps and sparse convolution  (K5, K10, K11, K13)
 *
 * Replaces  spnn.Conv3d (k3 s1 / k2 s2 / transposed / k1)   models/modules.py:19-64,90-122,181
 *           spconv SubMConv3d (k1 / k3, bias)               models/modules.py:252,444
 * nbr is int32[K][n]: nbr[k][i] = row of the voxel at coords[i] + offset_k * stride in the set the
 * table was built on, or -1.  ksize 3: K = 27, offsets in {-1,0,1}^3 with x fastest;
 * ksize 2: K = 8, offsets in {0,1}^3 with z fastest (k = 4 bx + 2 by + bz).
 *   stride-1 k3 conv on a set S:   table(S), coords = S, stride = tensor stride of S
 *   k2s2 down-conv S -> T:         table(S), coords = T (= unique of S at quantum 2 ts), stride = ts
 *   its transpose T -> S:          eprecon_transpose_map_async(S, parent = inverse from the unique call)
 * ------------------------------------------------------------------------------------------ */
int eprecon_kernel_map_async(const void *table, uint32_t capacity, const int32_t *coords, int64_t n,
                             int ksize, int stride, int32_t *nbr, void *stream);
int eprecon_transpose_map_async(const int32_t *fine_coords, int64_t n, const int32_t *parent,
                                int fine_stride, int32_t *up_map, void *stream);
/*
 * out[i, 0:cout] (op)= bias + sum_k x[nbr[k][i], 0:cin] @ weight[k]      weight f32[kvol][cin][cout]
 * x / out are row-major with leading dimensions ld_x / ld_out (so a layer can read from / write
 * into a channel slice of a wider buffer = torchsparse.cat for free).  nbr == NULL means the
 * identity map (kvol must be 1: a per-voxel linear layer).  relu != 0 fuses ReLU; accumulate != 0
 * adds to the existing contents of out.  fp32 MFMA, deterministic.
 */
int eprecon_sparse_conv_async(const float *x, int64_t n_in, int ld_x, const int32_t *nbr, int kvol,
                              int64_t n_out, const float *weight, int cin, int cout, const float *bias,
                              float *out, int ld_out, int relu, int accumulate, void *stream);
/*
 * The same contraction with the epilogues the reference wires around its convolutions:
 *   v = conv + bias [+ out]; [v = relu(v)]; [v += residual];  out = v
 * (residual: BN(x + ReLU(conv(x))) blocks, models/modules.py:385-399), and optionally the first
 * half of the train-mode BatchNorm that follows (models/modules.py:372-383 and every
 * spnn.BatchNorm): bn_partial f32[ceil(n_out/128)][3][cout] receives per-workgroup
 * (count, mean, M2) summaries of the stored values (eprecon_conv_bn_partial_bytes bytes), to be
 * finished by eprecon_batchnorm_apply_partials_async without re-reading the tensor for statistics.
 */
size_t eprecon_conv_bn_partial_bytes(int64_t n_out, int cout);
int eprecon_sparse_conv_fused_async(const float *x, int64_t n_in, int ld_x, const int32_t *nbr, int kvol,
                                    int64_t n_out, const float *weight, int cin, int cout,
                                    const float *bias, const float *residual, int ld_res, float *out,
                                    int ld_out, int relu, int accumulate, float *bn_partial, void *stream);
/*
 * Descriptor form with the remaining fusions of a conv -> BatchNorm(train) -> [ReLU] chain, so that
 * one launch per layer is left (the 2D fusion stack of models/occupancy_initialization.py:22-58 is
 * ~32 such layers and launch-bound otherwise):
 *   in_scale / in_shift [cin]   BatchNorm of the producer layer applied while gathering:
 *                               a = [relu]( x * in_scale[c] + in_shift[c] ), zero padding stays 0
 *   res_scale / res_shift [cout] the same for the residual operand
 *   bn_partial                  per-workgroup (count, mean, M2) summaries of THIS layer's stored output:
 *                               eprecon_batchnorm_finalize_affine_async turns them into (scale, shift)
 * The stored tensor is the un-normalised conv output; consumers apply (scale, shift) on load, or
 * eprecon_affine_rows_async materialises it.  (Round 3 also finished the statistics INSIDE the launch — write-through
 * summaries + arrival counters; parity-green and slower than the 7 us finalize launch, removed in round 4: DESIGN.md 7c.)
 */
typedef struct eprecon_conv_desc {
    const float *x; int64_t n_in; int ld_x;
    const int32_t *nbr; int kvol; int64_t n_out;
    const float *weight; int cin; int cout;
    const float *bias;
    const float *residual; int ld_res;
    float *out; int ld_out;
    int relu; int accumulate;
    const float *in_scale; const float *in_shift; int in_relu;
    const float *res_scale; const float *res_shift; int res_relu;
    float *bn_partial;
    /* row-wise LayerNorm over the cout channels after bias / ReLU / residual (the spconv + LayerNorm
     * blocks of models/modules.py:447-452,473-482, models/occupancy_initialization.py:141-169):
     * out = [relu]( LN(v) * ln_gamma + ln_beta ); cout <= 128, excludes bn_partial / accumulate */
    int ln; const float *ln_gamma; const float *ln_beta; float ln_eps; int ln_post_relu;
    /* dense 2D 3x3 'same' convolution over img_maps images of img_h x img_w pixel rows (kvol == 9,
     * n_out == img_maps * img_h * img_w): narrow layers then run as an implicit GEMM on image tiles
     * (halo tile + all nine weight matrices in LDS, no kernel map); nbr is still required as the
     * fallback for shapes the tile kernel does not take */
    int img_h; int img_w; int img_maps;
    /* dense-grid form of a 3x3x3 stride-1 convolution (kvol == 27) for voxel sets that fill most of their bounding
     * grid (the submanifold stack of models/occupancy_initialization.py:131-174 on the dense 48^3 grid): vox_rank
     * int32[grid_x*grid_y*grid_z] from eprecon_grid_rank_async maps a grid cell to its voxel's row (-1: none).
     * Layers the tile kernels take (cin % 4 == 0, cin <= 64: cout == 1, or cout <= 32 with cin % 16 == 0 and
     * packed_weight16) run as an implicit GEMM on cell tiles with the halo rows staged once in LDS: no kernel map, nbr
     * may be NULL.  Other shapes fall back to nbr (EPRECON_ERR_ARG when it is NULL). */
    const int32_t *vox_rank; int grid_x; int grid_y; int grid_z;
    /* `weight` in the operand order of v_mfma_f32_32x32x2_f32 (eprecon_conv_pack_weight_async): lets the short-list
     * (split-K) and cross-workgroup (medium lists, wide channels) kernels read their B operands without LDS staging */
    const float *packed_weight;
    /* the same weights in the operand order of the 16x16x4 MFMA kernels (eprecon_conv_pack_weight16_async, cout <= 64):
     *  - with vox_rank: the 16-row tile kernel, which takes cout <= 32 with cin a multiple of 16 (EPRECON_CONV_DENSE3D >= 2);
     *  - with nbr (kvol 27, or 9 for a pixel map): the direct gather kernel for lists the short-list rule does not take
     *    (operands straight from L2 into the MFMAs, no LDS staging; EPRECON_CONV_DIRECT=0 switches it off).  NULL keeps the
     *    launch on the 32x32x2 kernels.
     * Results equal those kernels' within fp32 round-off (four input channels per MFMA instead of two: another summation
     * order); the torchsparse / spconv layers this replaces (models/modules.py:15-72, 178-222) make no ordering promise. */
    const float *packed_weight16;
    /* scratch for kernels that reduce partial sums across workgroups (medium lists with wide channels, 3x3x3, with
     * packed_weight): eprecon_conv_desc_workspace_bytes(desc) bytes (0: none needed); without it those shapes stay on the
     * short-list kernel */
    void *workspace; size_t workspace_bytes;
    /* BatchNorm of the OUTPUT without a finalize launch (round 6; replaces bn_partial + eprecon_batchnorm_finalize_affine_async
     * between two layers, models/modules.py:15-29,313-399): the launch adds per-channel sums of the stored values to an
     * accumulator block with order-independent 64-bit integer atomics (exact fixed-point limbs: deterministic), and the
     * consumer of this layer's output names the block as in_acc and finishes it in its prologue.  A block of `ld` channels is
     * int64[eprecon_bn_acc_words(ld)]: int64[8][ld][7] sums, ZEROED by the caller before the producer runs, followed by
     * float[2][ld] (gamma, beta: the producer copies bn_gamma / bn_beta there; NULL = 1 / 0); bn_acc_c0 = first channel of
     * this layer inside the block (layers writing channel slices of one concatenated activation share a block).  Only for
     * launches eprecon_conv_desc_takes_bn_acc() accepts (no ln / accumulate, not the dense-grid 3D tile kernels). */
    long long *bn_acc; int bn_acc_ld; int bn_acc_c0; const float *bn_gamma; const float *bn_beta;
    /* ... and of the INPUT: the block the producer(s) of x filled, instead of in_scale / in_shift (in_relu still applies);
     * in_affine_scratch float[2 * cin]: where the library finishes the block with a launch of its own for the kernels that do
     * not do it in their prologue */
    const long long *in_acc; int in_acc_ld; int in_acc_c0; float in_eps; float *in_affine_scratch;
} eprecon_conv_desc;
/* int64 words of an accumulator block of `ld` channels (sums + the two parameter vectors) */
size_t eprecon_bn_acc_words(int ld);
/* 1 when the launch described by desc can produce into bn_acc (ask before setting it) */
int eprecon_conv_desc_takes_bn_acc(const eprecon_conv_desc *desc);
/* a block -> (scale, shift) vectors of `channels` channels from acc_c0 on: the stand-alone finish */
int eprecon_batchnorm_acc_affine_async(const long long *acc, int acc_ld, int acc_c0, int channels, float eps, float *scale_out,
                                       float *shift_out, void *stream);
/* out[i, c] = [relu]( x[i, c] * scale[c] + shift[c] ) with (scale, shift) finished from a block by every workgroup (channels <= 512) */
int eprecon_affine_rows_acc_async(const float *x, int64_t n, int channels, int ld_x, const long long *acc, int acc_ld, int acc_c0,
                                  float eps, int relu, float *out, int ld_out, void *stream);
int eprecon_conv_desc_async(const eprecon_conv_desc *desc, void *stream);
size_t eprecon_conv_desc_workspace_bytes(const eprecon_conv_desc *desc);
/* number of bn_partial rows the launch described by desc writes (nblk of the finalize call) */
int64_t eprecon_conv_desc_partial_rows(const eprecon_conv_desc *desc);
/* producer-side summaries partial f32[nblk][3][channels] -> the BatchNorm in affine form */
int eprecon_batchnorm_finalize_affine_async(const float *partial, int64_t nblk, int channels, const float *gamma,
                                            const float *beta, float eps, float *scale_out, float *shift_out,
                                            void *stream);
/* out[i, c] = [relu]( x[i, c] * scale[c] + shift[c] ); out may alias x */
int eprecon_affine_rows_async(const float *x, int64_t n, int channels, int ld_x, const float *scale,
                              const float *shift, int relu, float *out, int ld_out, void *stream);
/* out[i, c] = [relu]( x[i, c] * scale[c] + shift[c] + residual[i, c] ): the tail of the residual blocks
 * (models/modules.py:46-72) from a producer-finished BatchNorm; residual may be NULL; out may alias x */
int eprecon_affine_rows_res_async(const float *x, int64_t n, int channels, int ld_x, const float *scale,
                                  const float *shift, const float *residual, int ld_res, int relu, float *out,
                                  int ld_out, void *stream);
/*
 * Kernel map of a dense 2D 'same' convolution (odd ksize) over `maps` images of height x width
 * pixels stored as rows [maps][height][width] of a channels-last tensor:
 * nbr int32[ksize*ksize][maps*height*width], offset index ky * ksize + kx, -1 = zero padding.
 * With it nn.Conv2d(padding="same") of models/modules.py:313-399 / models/occupancy_initialization.py:22-31
 * runs on eprecon_sparse_conv_fused_async (weight re-laid out to [ky*ks+kx][cin][cout]).
 */
int eprecon_pixel_map_async(int maps, int height, int width, int ksize, int32_t *nbr, void *stream);
/*
 * Rank volume of a voxel set that lives on a dense grid (coords int32[n,4] (b,x,y,z), multiples of `stride`,
 * 0 <= x / stride < grid_x ...): rank int32[grid_x*grid_y*grid_z + 1], z fastest; rank[cell] = row of the voxel or -1,
 * the last element counts voxels off the grid (0 for a valid set).  Replaces the hash grid + 27-offset kernel map
 * for the dense-grid convolution form of eprecon_conv_desc (the role spconv's indice pairs play at
 * models/modules.py:260-271 for the reference).
 */
int eprecon_grid_rank_async(const int32_t *coords, int64_t n, int stride, int grid_x, int grid_y, int grid_z, int32_t *rank,
                            void *stream);
/* weight f32[kvol][cin][cout] -> the operand order of the dense-grid kernel (eprecon_conv_pack_weight_floats floats) */
size_t eprecon_conv_pack_weight_floats(int kvol, int cin, int cout);
int eprecon_conv_pack_weight_async(const float *weight, int kvol, int cin, int cout, float *packed, void *stream);
/* ... -> the operand order of the 16x16x4 MFMA kernels, cout <= 64 (0 floats / EPRECON_ERR_ARG beyond) */
size_t eprecon_conv_pack_weight16_floats(int kvol, int cin, int cout);
int eprecon_conv_pack_weight16_async(const float *weight, int kvol, int cin, int cout, float *packed, void *stream);
/* Many packings in one launch: `jobs` is an array of njobs (<= 65,535) descriptors in DEVICE memory; kind 0 = the order of
 * eprecon_conv_pack_weight_async, 1 = of eprecon_conv_pack_weight16_async (cout <= 64); `packed` sized by the matching
 * *_floats call.  An optimisation step (main.py:297-313: optimizer.step()) changes every weight, so the operand-order copies
 * of every spnn.Conv3d (models/modules.py:15-72, 178-222) are rebuilt once per step: one launch instead of one per layer. */
typedef struct eprecon_pack_job {
    const float *weight; float *packed;
    int32_t kvol; int32_t cin; int32_t cout; int32_t kind;
} eprecon_pack_job;
int eprecon_conv_pack_many_async(const eprecon_pack_job *jobs, int njobs, void *stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation epilogues  (K12)
 *
 * Replaces  spnn.BatchNorm / nn.BatchNorm1d in TRAIN mode (batch statistics over all active
 *           voxels; the reference tests in train mode, main.py:357)   models/modules.py:22,41,54,60,65
 *           nn.LayerNorm with the ReLU / residual wiring of           models/modules.py:447-452,473-482
 *                                                                     models/occupancy_initialization.py:141-169
 * ------------------------------------------------------------------------------------------ */
size_t eprecon_batchnorm_workspace_bytes(int64_t n, int channels);
/* out = [relu]( (x - mean) / sqrt(var + eps) * gamma + beta [+ residual] ), biased variance;
 * out may alias x; mean_out / var_out optional f32[channels] */
int eprecon_batchnorm_train_async(const float *x, int64_t n, int channels, int ld_x, const float *gamma,
                                  const float *beta, float eps, const float *residual, int ld_res,
                                  int relu, float *out, int ld_out, float *mean_out, float *var_out,
                                  void *workspace, size_t workspace_bytes, void *stream);
/* second half of the same BatchNorm from producer-side summaries partial f32[nblk][3][channels]
 * (count, mean, M2 per block, merged in block order) */
size_t eprecon_batchnorm_apply_workspace_bytes(int channels);
int eprecon_batchnorm_apply_partials_async(const float *x, int64_t n, int channels, int ld_x,
                                           const float *partial, int64_t nblk, const float *gamma,
                                           const float *beta, float eps, const float *residual,
                                           int ld_res, int relu, float *out, int ld_out, float *mean_out,
                                           float *var_out, void *workspace, size_t workspace_bytes,
                                           void *stream);
/* the same with a residual operand that carries a pending BatchNorm of its own in affine form (the 1x1 skip convolution +
 * BatchNorm of a residual block, models/modules.py:57-65,71): residual' = residual * res_scale + res_shift on load */
int eprecon_batchnorm_apply_partials_res_async(const float *x, int64_t n, int channels, int ld_x, const float *partial,
                                               int64_t nblk, const float *gamma, const float *beta, float eps,
                                               const float *residual, int ld_res, const float *res_scale,
                                               const float *res_shift, int relu, float *out, int ld_out, void *workspace,
                                               size_t workspace_bytes, void *stream);
/* per row: t = x; if pre_relu t = relu(t); if residual t += residual; y = LN(t) * gamma + beta;
 * if post_relu y = relu(y).  out may alias x. */
int eprecon_rowwise_layernorm_async(const float *x, int64_t n, int channels, int ld_x,
                                    const float *residual, int ld_res, const float *gamma,
                                    const float *beta, float eps, int pre_relu, int post_relu,
                                    float *out, int ld_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Dense-grid helpers of the coarse-to-fine loop  (K14, K16)
 * ------------------------------------------------------------------------------------------ */
/*
 * Initialisation -> coarse selection (models/neucon_network.py:264,298-318): per batch element,
 * mark the (dim^3) coarse cells that contain a voxel with sigmoid(logit) > threshold (the 2^3
 * max-pool of the 48^3 volume when cell = 4 and the voxels sit on the interval-2 grid), erode with
 * a zero-padded 3^3 box, dilate twice, and emit the surviving cells in raster order as
 * (b, cell*x, cell*y, cell*z).  coords int32[n,4] are the valid voxels the logits belong to.
 * out_coords int32[batch*dim^3, 4]; n_out_dev int32[1 + batch] ([0] total, [1+b] per batch).
 */
size_t eprecon_init_select_workspace_bytes(int batch, int dim);
int eprecon_init_select_async(const float *logit, const int32_t *coords, int64_t n, float threshold,
                              int batch, int dim, int cell, int32_t *out_coords, int32_t *n_out_dev,
                              void *workspace, size_t workspace_bytes, void *stream);
/*
 * Sparsify for the next stage (models/neucon_network.py:454-507) in one call: occupancy = occ > threshold; the kept rows of
 * coords / tsdf / occ / feat_all compacted in row order (torch.nonzero + index_select + cat of the reference) and the counts
 * its guards read: counts int32[1 + 2 * batch] = [kept rows, occupied per batch element ..., occupied with an occupied
 * target per batch element ...] (target uint8[n] or NULL = all ones).  Outputs are sized for n rows:
 * out_coords int32[n,4], out_tsdf / out_occ f32[n], out_all f32[n, c_all], out_feat f32[n, c_feat + 2] = [feat_all[:, :c_feat],
 * tsdf, occ] (the next stage's input).  The random sub-sampling branch (:477-484) is not part of it.
 */
size_t eprecon_sparsify_workspace_bytes(int64_t n);
int eprecon_sparsify_async(const float *occ, int ld_occ, float threshold, const unsigned char *target,
                           const int32_t *coords, const float *tsdf, int ld_tsdf, const float *feat_all, int ld_feat,
                           int c_all, int c_feat, int64_t n, int batch, int32_t *out_coords, float *out_tsdf,
                           float *out_occ, float *out_all, float *out_feat, int32_t *counts, void *workspace,
                           size_t workspace_bytes, void *stream);
/*
 * NeuConNet.upsample (models/neucon_network.py:193-214): up_coords int32[8n,4], up_feat f32[8n,C];
 * children of a voxel are consecutive, in the order 0, +x, +y, +z, +xy, +xz, +yz, +xyz (x `interval`).
 * channels == 0 expands the coordinates only; up_coords NULL (channels > 0) expands the features only.
 */
int eprecon_upsample_async(const float *feat, int ld_feat, const int32_t *coords, int64_t n, int channels,
                           int interval, float *up_feat, int32_t *up_coords, void *stream);

/* ------------------------------------------------------------------------------------------
 * Point <-> voxel transfers  (K8, K9) and aligned-camera coordinates (a7)
 *
 * Replaces  r_coords = W2AC[:3] . [c * vs + origin, 1]   models/neucon_network.py:387-398,
 *                                                         models/gru_fusion.py:332-337
 *           initial_voxelize / point_to_voxel / voxel_to_point   ops/torchsparse_utils.py:15-105
 *           (torchsparse F.spcount, F.spvoxelize, F.calc_ti_weights, F.spdevoxelize)
 * Points are f32[n,4] rows (x, y, z, batch) like torchsparse's PointTensor.C.
 * ------------------------------------------------------------------------------------------ */
int eprecon_aligned_coords_async(const int32_t *coords, int64_t n, const float *origin, int batch,
                                 float voxel_size, const float *world_to_aligned_camera, float *out_xyzb,
                                 void *stream);
/* scaled = (x/res, y/res, z/res, b) (IEEE division); voxel = (b, floor x', floor y', floor z') */
int eprecon_point_quantize_async(const float *points_xyzb, int64_t n, float resolution, float *scaled_xyzb,
                                 int32_t *voxel_bxyz, void *stream);
/*
 * The coordinate side of the NEXT SPVCNN pass, queued BEFORE the host has read how many rows a compaction kept
 * (models/neucon_network.py:454-507 keeps the occupied rows, :193-214 expands them to 8 children, :387-398 maps them to the
 * aligned camera frame, ops/torchsparse_utils.py:15-19 scales and floors them): src_coords int32[cap_src,4] with the live row
 * count in *n_src_dev; children != 0: every row -> its 8 children (interval of the new level) in up_coords int32[8 cap_src,4];
 * r_coords / scaled_xyzb f32[cap_pts,4], voxel_bxyz int32[cap_pts,4], *n_points_dev = live points.  One launch sized by the
 * capacity; the unique-voxel numbering (eprecon_unique_coords_dn_async) then runs on *n_points_dev, and the level's size read
 * rides on the read of the compaction's own counts (one blocking read instead of two).  Same arithmetic as
 * eprecon_upsample_async + eprecon_aligned_coords_async + eprecon_point_quantize_async.
 */
int eprecon_spvcnn_points_dn_async(const int32_t *src_coords, int64_t cap_src, const int32_t *n_src_dev, int children,
                                   int interval, const float *origin, int batch, float voxel_size,
                                   const float *world_to_aligned_camera, float resolution, int32_t *up_coords, float *r_coords,
                                   float *scaled_xyzb, int32_t *voxel_bxyz, int32_t *n_points_dev, void *stream);
/* ... with the point count on the device (see eprecon_unique_coords_dn_async) */
int eprecon_point_quantize_dn_async(const float *points_xyzb, int64_t n_cap, const int32_t *n_dev, float resolution,
                                    float *scaled_xyzb, int32_t *voxel_bxyz, void *stream);
/* CSR lists of the points of each voxel: idx int32[n] in [-1, m) -> offsets int32[m+1],
 * order int32[n] (points of voxel v = order[offsets[v] : offsets[v+1]], ascending point index) */
size_t eprecon_segment_workspace_bytes(int64_t n, int64_t m);
int eprecon_segment_lists_async(const int32_t *idx, int64_t n, int64_t m, int32_t *offsets, int32_t *order,
                                void *workspace, size_t workspace_bytes, void *stream);
/* out[v] = mean of feat[p] over the points p of voxel v (0 for empty voxels) — scatter-mean
 * without float atomics */
int eprecon_segment_mean_async(const float *feat, int ld_feat, const int32_t *offsets, const int32_t *order,
                               int64_t m, int channels, float *out, int ld_out, void *stream);
/*
 * Voxel ORDER of the reference.  torchsparse numbers the voxels of initial_voxelize by ascending
 * F.sphash (`torch.unique(pc_hash)`, ops/torchsparse_utils.py:19-21); nothing depends on that order
 * except ConvGRU's second gate convolution (models/modules.py:216-217), which devoxelises with the
 * corner indices cached by the first voxelisation of the same PointTensor into the second voxel
 * set (ops/torchsparse_utils.py:70-71,97-99).  eprecon_sphash_async returns torchsparse's 60-bit
 * FNV-1a hash of every (b,x,y,z) row (hashed in x,y,z,b order); the caller sorts it.
 * eprecon_remap_index_async rewrites cached indices:  out = perm_new[rank_old[idx]]  (-1 stays -1;
 * a rank >= m_new, an out-of-bounds read in the reference, becomes -1).
 */
int eprecon_sphash_async(const int32_t *coords, int64_t n, int64_t *out_hash, void *stream);
/* the whole order in one call (csrc/hash_order.hip): perm_out[k] = row of the voxel with the k-th smallest hash (hashes of
 * distinct voxels are distinct in practice; equal hashes keep their row order), rank_out = the inverse permutation.
 * workspace: eprecon_sphash_order_workspace_bytes(n). */
size_t eprecon_sphash_order_workspace_bytes(int64_t n);
int eprecon_sphash_order_async(const int32_t *coords, int64_t n, int32_t *perm_out, int32_t *rank_out, void *workspace,
                               size_t workspace_bytes, void *stream);
int eprecon_remap_index_async(const int32_t *idx, int64_t n, const int32_t *rank_old, const int32_t *perm_new,
                              int64_t m_new, int32_t *out, void *stream);
/* 8-corner indices int32[n,8] and renormalised trilinear weights f32[n,8] of points (in voxel
 * units) against the voxel set the table was built on, at tensor stride `stride` */
int eprecon_trilinear_map_async(const void *table, uint32_t capacity, const float *points_xyzb, int64_t n,
                                int stride, int32_t *idx8, float *weight8, void *stream);
/* out[i] (+)= sum_k weight8[i,k] * voxel_feat[idx8[i,k]] */
int eprecon_devoxelize_async(const float *voxel_feat, int ld_feat, const int32_t *idx8, const float *weight8,
                             int64_t n, int channels, float *out, int ld_out, int accumulate, void *stream);

/*
 * The tail of SConv3d inside ConvGRU (models/modules.py:193-196,214-221) in one launch:
 * v = devoxelise(voxel_feat) + skip  (skip = the point-wise Linear of SConv3d), then
 *   mode 1: out = sigmoid(v)                    mode 2: out = sigmoid(v) * h
 *   mode 3: out = (1 - zgate) * h + zgate * tanh(v)
 * out may be a column slice of the [r*h, x] concat buffer.
 */
#define EPRECON_GATE_SIGMOID 1
#define EPRECON_GATE_SIGMOID_MUL 2
#define EPRECON_GATE_GRU_MIX 3
int eprecon_devoxelize_gate_async(const float *voxel_feat, int ld_feat, const int32_t *idx8, const float *weight8,
                                  int64_t n, int channels, const float *skip, int ld_skip, int mode, const float *h,
                                  int ld_h, const float *zgate, int ld_z, float *out, int ld_out, void *stream);

/* ... and, in the same launch, tail_dst[i, 0:tail_channels] = tail_src[i, 0:tail_channels]: with mode 2 writing r * h into the
 * first half of the [r*h, x] buffer (models/modules.py:218) the x half is copied alongside, so the buffer needs no clone of [h, x] */
int eprecon_devoxelize_gate_tail_async(const float *voxel_feat, int ld_feat, const int32_t *idx8, const float *weight8,
                                       int64_t n, int channels, const float *skip, int ld_skip, int mode, const float *h,
                                       int ld_h, const float *zgate, int ld_z, float *out, int ld_out, const float *tail_src,
                                       int ld_tail_src, float *tail_dst, int ld_tail_dst, int tail_channels, void *stream);

/* ------------------------------------------------------------------------------------------
 * GRU-fusion union  (K15)
 *
 * Replaces  GRUFusion.convert2dense + the dense gathers    models/gru_fusion.py:67-114,321-326
 *           sparse_to_dense_channel                         utils.py:176-180
 * Union, in raster order of the local dim^3 grid, of (a) the current fragment's voxels
 * (cur_coords int32[n_cur,4] (b,x,y,z) finest units of ONE batch element, divided by `interval`)
 * and (b) the global-map voxels (glob_coords int32[n_glob,3], scene grid units of this scale)
 * shifted by -relative_origin that land inside [0,dim)^3.  A voxel is active when its current or
 * its global feature row has a non-zero channel (activity_mode 0, the reference's
 * `(volume != 0).any(-1)`) or a channel with |v| < 1 (activity_mode 1, TSDF direct substitution).
 *   updated  int32[n_out,3]   src_cur / src_glob int32[n_out] (row or -1)   glob_valid u8[n_glob]
 * ------------------------------------------------------------------------------------------ */
size_t eprecon_fbv_union_workspace_bytes(int dim);
int eprecon_fbv_union_async(const int32_t *cur_coords, const float *cur_feat, int64_t n_cur, int ld_cur,
                            const int32_t *glob_coords, const float *glob_feat, int64_t n_glob, int ld_glob,
                            int channels, int dim, int interval, int activity_mode,
                            const int32_t *relative_origin_host, int32_t *updated, int32_t *src_cur, int32_t *src_glob, uint8_t *glob_valid,
                            int32_t *n_out_dev, void *workspace, size_t workspace_bytes, void *stream);
/* out[i] = src[i] >= 0 ? feat[src[i]] : fill   (rows of `channels` floats) */
int eprecon_gather_rows_async(const float *feat, int ld_feat, const int32_t *src, int64_t n, int channels,
                              float fill, float *out, int ld_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Persistent global map of GRU fusion behind an opaque handle  (K15, SURVEY.md 8b "Ownership")
 *
 * Replaces  GRUFusion.global_volume / target_tsdf_volume state       models/gru_fusion.py:31-38,59-65
 *           convert2dense (crop to the FBV, union, gathers)          models/gru_fusion.py:67-114,321-326
 *           update_map  (map = map[outside FBV] ++ fused rows)       models/gru_fusion.py:195-215
 * A handle owns its device memory (rows int32[M,3] scene-grid coordinates of one scale + f32[M,C]
 * features, ping-pong buffers with geometric growth) and is the ONE stateful object of the ABI:
 * create once per scale, reset on scene change (models/gru_fusion.py:283-286), destroy at the end.
 * One fragment = crop_union -> gather(s) -> [caller fuses] -> update.  Row order after update is the
 * reference's: rows outside the FBV in their old order, then the union voxels in raster order.
 * Calls on one handle must be issued from one thread; crop_union / target_fuse synchronise `stream`
 * (the reference's torch.nonzero synchronises at the same point), the others are stream-ordered.
 * ------------------------------------------------------------------------------------------ */
int eprecon_map_create(int channels, void **out_handle);
int eprecon_map_destroy(void *handle);
int eprecon_map_reset(void *handle);                 /* size := 0, memory kept */
int64_t eprecon_map_size(const void *handle);        /* rows (host-side count) */
int eprecon_map_channels(const void *handle);
/* copy the rows out (coords int32[size,3], feats f32[size,C], both dense) / replace the contents */
int eprecon_map_export_async(const void *handle, int32_t *coords_out, float *feats_out, void *stream);
int eprecon_map_import_async(void *handle, const int32_t *coords, const float *feats, int64_t n, void *stream);
/*
 * Union, in raster order of the local dim^3 grid, of the current fragment's voxels (as in
 * eprecon_fbv_union_async) and the map rows that fall inside the FBV at relative_origin_host.
 * Outputs have capacity min(dim^3, n_cur + size) rows; counts_host[0] = n_out, [1] = map rows inside.
 * The rows inside are remembered by the handle for the next eprecon_map_update_async.  Blocking.
 */
int eprecon_map_crop_union(void *handle, const int32_t *cur_coords, const float *cur_feat, int64_t n_cur, int ld_cur,
                           int dim, int interval, int activity_mode, const int32_t *relative_origin_host,
                           int32_t *updated, int32_t *src_cur, int32_t *src_glob, int64_t *counts_host, void *stream);
/* out[i, 0:channels] = src_glob[i] >= 0 ? map.feats[src_glob[i], col0 : col0 + channels] : fill */
int eprecon_map_gather_async(const void *handle, const int32_t *src_glob, int64_t n, int col0, int channels, float fill,
                             float *out, int ld_out, void *stream);
/* update_map: drop the rows inside the last crop's FBV, append (updated + relative origin, values) */
int eprecon_map_update_async(void *handle, const int32_t *updated, int64_t n, const float *values, int ld_values,
                             void *stream);
/*
 * Ground-truth twin (a 1-channel map; the reference's test path reads the targets,
 * models/neucon_network.py:488): dense volume default 1 <- map rows inside the FBV <- the fragment's
 * ground truth where occ_gt (tsdf_gt f32[dim^3], occ_gt u8[dim^3]); tsdf_target_out[i] = volume at
 * updated[i]; then map = map[outside] ++ raster-ordered cells with |v| < 1.  Blocking.
 */
int eprecon_map_target_fuse(void *handle, const float *tsdf_gt, const uint8_t *occ_gt, int dim,
                            const int32_t *relative_origin_host, const int32_t *updated, int64_t n, float *tsdf_target_out,
                            void *stream);
/*
 * One GRU-fusion level as ONE stream-ordered call with device-side counts  (models/gru_fusion.py:259-345 up to the ConvGRUs:
 * convert2dense + the gathers :321-326, the ground-truth twin :99-113, the fragment's points and their aligned-camera
 * coordinates :328-337, and the coordinate side of the two voxelisations the six SConv3d of the two ConvGRUs share,
 * ops/torchsparse_utils.py:15-35).  Replaces eprecon_map_crop_union (blocking) + 4 gathers + eprecon_map_target_fuse (blocking) +
 * eprecon_aligned_coords_async + 2 x (eprecon_point_quantize_async, eprecon_unique_coords_async + a host read each): nothing
 * here waits for the device.  Every output has `capacity` rows (eprecon_gru_stage_capacity: min(dim^3, n_cur + map rows));
 * counts int32[8] (device) = [0] union voxels n_u, [1] map rows outside the FBV, [2] / [3] unique voxels of the first / second
 * voxelisation, [4] ground-truth cells appended, [5] ground-truth rows kept, [6] / [7] status words of table1 / table2.
 * The caller reads counts back ONCE, hands them to eprecon_gru_stage_commit_async (which finishes the ground-truth twin's
 * update and arms eprecon_map_update_async) and slices its buffers; kernel maps / point lists / corner tables are then built
 * on the exact sizes.  The second voxelisation is that of the ALREADY SCALED points (models/modules.py:216-217 with the
 * in-place z.C of ops/torchsparse_utils.py:33).
 */
typedef struct eprecon_gru_stage_desc {
    void *map;                    /* feature map of this scale (C channels) */
    void *target_map;             /* ground-truth twin (1 channel) or NULL */
    const int32_t *cur_coords;    /* int32[n_cur,4] (b,x,y,z), finest units, ONE batch element */
    const float *cur_feat;        /* f32[n_cur, ld_cur] */
    int64_t n_cur; int ld_cur;
    int dim; int interval; int activity_mode;
    int32_t rel[3];               /* relative origin of the fragment volume in the scene grid of this scale */
    const float *tsdf_gt; const uint8_t *occ_gt;   /* [dim^3] ground truth of the fragment, with target_map */
    const float *origin;          /* f32[3] (device): vol_origin_partial of this batch element */
    const float *w2ac;            /* f32[16] (device): world_to_aligned_camera of this batch element */
    float voxel_size;             /* finest voxel size */
    float resolution;             /* what the SConv3d divide the point coordinates by (vres / pres) */
    int ch_voxel;                 /* channels [0, ch_voxel) feed the voxel ConvGRU, [ch_voxel, C) the image ConvGRU */
    int batch_index;              /* batch column of out_coords */
    int64_t capacity;
    int32_t *updated;             /* int32[cap,3] union cells, raster order */
    int32_t *out_coords;          /* int32[cap,4] (batch_index, cell * interval) */
    float *r_coords;              /* f32[cap,4] aligned-camera (x,y,z,0) */
    float *hx_voxel;              /* f32[cap, 2 ch_voxel]      [h | x]: the map's row, the fragment's row (zeros where absent) */
    float *hx_image;              /* f32[cap, 2 (C - ch_voxel)] */
    float *tsdf_target;           /* f32[cap] ground-truth TSDF at the union cells, with target_map */
    float *scaled1; int32_t *vox1; int32_t *inverse1; int32_t *uniq1; void *table1;   /* f32[cap,4], int32[cap,4], [cap], [cap,4] */
    float *scaled2; int32_t *vox2; int32_t *inverse2; int32_t *uniq2; void *table2;
    uint32_t table_capacity;      /* eprecon_hash_capacity(capacity) */
    int32_t *counts;              /* int32[8] (device) */
    void *workspace; size_t workspace_bytes;   /* eprecon_gru_stage_workspace_bytes(capacity) */
} eprecon_gru_stage_desc;
int64_t eprecon_gru_stage_capacity(const void *map, int64_t n_cur, int dim);
size_t eprecon_gru_stage_workspace_bytes(int64_t capacity);
int eprecon_gru_stage_begin_async(const eprecon_gru_stage_desc *desc, void *stream);
/* counts_host int32[8]: the host copy of desc->counts */
int eprecon_gru_stage_commit_async(void *map, void *target_map, const int32_t *counts_host, void *stream);
/*
 * Multi-GPU boundary exchange on the handle (the schedule of eprecon_amd/distributed.py, which emulates the
 * sequential map updates of models/gru_fusion.py:195-215,275 across ranks).  Every row carries a stamp: 0 unknown,
 * +(fragment + 1) fused by THIS rank, -(fragment + 1) received.  eprecon_map_set_fragment: the global fragment index
 * eprecon_map_update_async stamps its appended rows with (-1: none).  eprecon_map_stamps_async: import / fill / export the
 * int32[size] stamps (in that order; NULL / 0 skips a step).
 *   select  flags the rows fused here that lie inside any box boxes_lo[b] + [0, dim)^3 with b != own_box (device int32[n_boxes,3],
 *           scene-grid units of this scale) and writes their number to count_out (device int32); asynchronous
 *   pack    writes the selected rows in map order as payload f32[n_rows][4 + channels]: (x, y, z, fragment) as int32 bit
 *           patterns, then the features; n_rows = the count select produced (read back by the caller with the other scales')
 *   merge   applies received payload rows that fall inside the local box: per cell the newest copy wins (atomicMax + one
 *           claim), it overwrites the local row when newer than it or is appended in payload order; received rows are
 *           stamped negative so they are not re-broadcast.  Blocking (one host read: the number of appended rows).
 */
int eprecon_map_set_fragment(void *handle, int fragment_index);
int eprecon_map_stamps_async(void *handle, int32_t *export_to, const int32_t *import_from, int fill_all, int32_t fill_value,
                             void *stream);
int eprecon_map_select_boundary_async(void *handle, const int32_t *boxes_lo, int n_boxes, int own_box, int dim,
                                      int32_t *count_out, void *stream);
int eprecon_map_pack_boundary_async(void *handle, float *payload, int64_t n_rows, void *stream);
int eprecon_map_merge_boundary(void *handle, const float *payload, int64_t n_rows, const int32_t *box_lo_host, int dim,
                               int64_t *n_added_host, void *stream);

/* ------------------------------------------------------------------------------------------
 * TSDF integration of depth frames  (the data-preparation side of the path)
 *
 * Replaces  TSDFVolumeTorch.integrate / integrate()     tools/tsdf_fusion/fusion.py:440-485,551-575
 *           (run on the CPU for every sample: datasets/transforms.py:286-297,375-387)
 *           and the in-tree PyCUDA kernel `integrate`   tools/tsdf_fusion/fusion.py:67-142
 * tsdf / weight f32[dims[0], dims[1], dims[2]] (x-major, like the reference's volumes; a fresh volume is
 * tsdf = 1, weight = 0) are updated in place with ALL n_views frames in one launch (per voxel the views
 * are applied in order, as n successive integrate() calls would).  depth f32[n_views, height, width]
 * (metres along the camera z axis, 0 = invalid); intr_host f32[n_views,3,3]; cam_host f32[n_views,4,4]:
 * variant 0 (TSDFVolumeTorch arithmetic) world->camera matrices (the reference's torch.inverse(cam_pose)),
 * variant 1 (the PyCUDA kernel's arithmetic) the camera poses themselves.  trunc = margin * voxel_size.
 * occ_out u8[cells] or NULL: |tsdf| < 0.999 and weight > 1 after the last view (datasets/transforms.py:295-297).
 * dims_host / origin_host / intr_host / cam_host are HOST pointers.
 * ------------------------------------------------------------------------------------------ */
#define EPRECON_TSDF_TORCH 0
#define EPRECON_TSDF_CUDA 1
int eprecon_tsdf_integrate_async(float *tsdf, float *weight, const int32_t *dims_host, const float *origin_host,
                                 float voxel_size, const float *depth, int n_views, int height, int width,
                                 const float *intr_host, const float *cam_host, float trunc, float obs_weight,
                                 int variant, uint8_t *occ_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Scene mesh extraction: marching cubes on the dense scene TSDF  (scene output path)
 *
 * Replaces  skimage.measure.marching_cubes(tsdf_vol, level=0) + the vertex label lookups of
 *           SaveScene.tsdf2mesh / tsdf_panoptic2mesh                       utils.py:225-241
 * so the dense volumes of GRUFusion.save_mesh (models/gru_fusion.py:217-257) become a mesh on the device.
 * volume f32[dx,dy,dz] (x-major).  Two phases: _count (blocking: one host read of the sizes, the scanned
 * offsets stay in `workspace`), then _emit_async into caller buffers: verts f32[nv,3] in voxel
 * coordinates (the caller applies voxel_size / origin like utils.py:228), normals f32[nv,3] (unit field
 * gradient) or NULL, faces int32[nt,3] wound along the gradient; vert_label_a/b int32[nv]: labels of the
 * voxel nearest to each vertex from two optional int32 volumes (semantic / instance).
 * Vertices are the sign-change points any marching-cubes variant yields; the triangulation table is
 * generated from one rule (csrc/marching_cubes.hip) — skimage's Lewiner tables are not reproduced.
 * ------------------------------------------------------------------------------------------ */
int eprecon_marching_cubes_table(int8_t *out_host /* [256][16] */);
size_t eprecon_marching_cubes_workspace_bytes(int dx, int dy, int dz);
int eprecon_marching_cubes_count(const float *volume, int dx, int dy, int dz, float level, int64_t *counts_host,
                                 void *workspace, size_t workspace_bytes, void *stream);
int eprecon_marching_cubes_emit_async(const float *volume, int dx, int dy, int dz, float level, float *verts, float *normals,
                                      int32_t *faces, const int32_t *label_a, const int32_t *label_b, int32_t *vert_label_a,
                                      int32_t *vert_label_b, const void *workspace, void *stream);

/* ------------------------------------------------------------------------------------------
 * Nearest finest-level voxel  (K18)
 *
 * Replaces  torch.cdist + argmin(dim=1)                 models/mask3dformer.py:361-367
 * out_index[i] = row of the reference voxel (ref_coords int32[m,4] (b,x,y,z), hashed in `table`)
 * nearest to query_coords[i] (a voxel of a coarser level, multiple of `quantum`) in exact integer
 * Euclidean distance, smallest row on ties, same batch element; -1 when that batch element has no
 * reference voxel.
 * ------------------------------------------------------------------------------------------ */
int eprecon_nearest_voxel_async(const void *table, uint32_t capacity, const int32_t *ref_coords,
                                int64_t m, const int32_t *query_coords, int64_t n, int quantum,
                                int32_t *out_index, void *stream);

/* ------------------------------------------------------------------------------------------
 * Voxel side of the mask-transformer decoder  (K19)
 *
 * Replaces  the per-level inputs of MultiScaleMaskedTransformerDecoder.forward      models/mask3dformer.py:346-357
 *           (PositionEmbeddingCoordsSine "fourier", normalize=True                   models/voxel_position_encoding.py:123-152)
 *           the masked cross-attention of a decoder layer                            models/mask3dformer.py:383-397
 *           with the attention mask of forward_prediction_heads                      models/mask3dformer.py:429-445
 *
 * eprecon_decoder_keys_async: coords int32 rows (x, y, z) with row pitch ld_coords ints, feats f32[n, ld_feats]:
 *   src[i, c] = feats[i, c] + level_embed[c];  keys[i, c] = src[i, c] + pos[i, c],
 *   pos = [sin(P), cos(P)], P = (2 pi * coords / extent) @ gauss_b   (gauss_b f32[3, channels / 2], extent_host f32[3] HOST)
 *
 * eprecon_masked_attention_async: out f32[H, Q, D] = softmax_over_allowed_keys(scale * q k^T) v per head; q element
 *   (h, i, d) at q[h * q_stride_head + i * q_stride_query + d] (the head-split VIEW of a [Q, H * D] in-projection is taken as is),
 *   k / v f32[n_keys, ld] rows whose H * D channels are the heads side by side (what nn.MultiheadAttention's in-projection
 *   produces).  Key n is BLOCKED for query i when sigmoid(mask_logits_t[row(n), i]) < 0.5 with row(n) = mask_rows[n]
 *   (NULL: n); mask_logits_t f32[n_mask_rows, ld_mask] is the TRANSPOSED mask-logit matrix (voxel-major).  A query whose
 *   mask blocks every key attends to all keys (models/mask3dformer.py:388).  mask_logits_t NULL: no mask.
 *   Shapes taken: D = 6, H even, Q * H / 2 <= 512, Q <= 255, k / v 16-byte aligned with ld % 4 == 0;
 *   EPRECON_ERR_UNSUPPORTED otherwise (callers keep the dense PyTorch path).  Deterministic (fixed merge order).
 * ------------------------------------------------------------------------------------------ */
int eprecon_decoder_keys_async(const int32_t *coords, int ld_coords, const float *feats, int ld_feats, const float *level_embed,
                               const float *gauss_b, const float *extent_host, int64_t n, int channels, float *src_out,
                               float *keys_out, void *stream);
size_t eprecon_masked_attention_workspace_bytes(int64_t n_keys, int n_queries, int n_heads, int head_dim);
int eprecon_masked_attention_async(const float *q, int q_stride_head, int q_stride_query, const float *k, int ld_k, const float *v, int ld_v, int64_t n_keys,
                                   const float *mask_logits_t, int ld_mask, const int32_t *mask_rows, int64_t n_mask_rows,
                                   int n_queries, int n_heads, int head_dim, float scale, float *out, void *workspace,
                                   size_t workspace_bytes, void *stream);

/*
 * Query side of a decoder layer (static shape [Q, C]): out-projection + residual + LayerNorm of the cross-attention whose
 * per-head output is o_attn (eprecon_masked_attention_async), self-attention over the queries, FFN (post-norm blocks,
 * models/mask3dformer.py:33-196,399-427), the prediction head's class logits and mask embedding (:429-436) and the NEXT layer's
 * in-projected queries — two launches instead of ~25.  Every weight matrix is given TRANSPOSED, f32[in][out] row-major
 * (nn.Linear stores [out][in]); self_in_wt is nn.MultiheadAttention.in_proj_weight transposed: f32[C][3C] with the q | k | v
 * columns side by side.  workspace: 4 * Q * C floats.  Shapes taken: Q <= 128, C <= 64 with C % H == 0, H <= 8,
 * ffn_dim / mask_hidden <= 192, n_class_logits <= 64 (EPRECON_ERR_UNSUPPORTED otherwise).
 */
typedef struct eprecon_decoder_layer_desc {
    int n_queries; int channels; int n_heads; int ffn_dim; int n_class_logits; int mask_hidden;
    const float *o_attn;        /* f32[H][Q][C/H] */
    const float *state_in;      /* f32[Q][C] queries entering the layer */
    const float *query_pos;     /* f32[Q][C] */
    const float *cross_out_wt; const float *cross_out_b; const float *cross_ln_g; const float *cross_ln_b;
    const float *self_in_wt; const float *self_in_b; const float *self_out_wt; const float *self_out_b;
    const float *self_ln_g; const float *self_ln_b;
    const float *ffn1_wt; const float *ffn1_b; const float *ffn2_wt; const float *ffn2_b; const float *ffn_ln_g; const float *ffn_ln_b;
    const float *dec_ln_g; const float *dec_ln_b; const float *cls_wt; const float *cls_b;
    const float *m1_wt; const float *m1_b; const float *m2_wt; const float *m2_b; const float *m3_wt; const float *m3_b;
    const float *next_q_wt; const float *next_q_b;     /* the next layer's cross-attention q in-projection, or NULL */
    float ln_eps;
    float *state_out;           /* f32[Q][C] */
    float *cls_out;             /* f32[Q][n_class_logits] */
    float *mask_embed_out;      /* f32[Q][C] */
    float *next_q_out;          /* f32[Q][C] or NULL */
    float *workspace;
} eprecon_decoder_layer_desc;
int eprecon_decoder_query_side_async(const eprecon_decoder_layer_desc *desc, void *stream);

/*
 * panoptic_inference on the voxel side (models/mask3dformer.py:515-581 after the per-query softmax / keep decision):
 *   mask_logits f32[Q][ld] (the final head's pred_masks of one batch element), scores f32[Q], keep int32[Q] (label != 0 and
 *   score > threshold) -> owner_out int32[n]: the kept query with the largest score * sigmoid(logit) (first on ties, -1 when
 *   no query is kept), confident_out u8[n]: sigmoid(logit of the owner) >= 0.5, counts_out int32[3][Q]: voxels owned,
 *   voxels with sigmoid >= 0.5, both (zeroed by the call).  Integer atomics only: deterministic.
 * eprecon_panoptic_assign_async: seg[v] = idmap[owner[v]] where the owner is confident, else 0 (idmap int32[Q]: the segment id
 *   the host gave the query, 0 for rejected ones).
 */
int eprecon_panoptic_stats_async(const float *mask_logits, int64_t ld, const float *scores, const int32_t *keep, int n_queries,
                                 int64_t n, int32_t *owner_out, uint8_t *confident_out, int32_t *counts_out, void *stream);
int eprecon_panoptic_assign_async(const int32_t *owner, const uint8_t *confident, const int32_t *idmap, int64_t n, int32_t *seg_out,
                                  void *stream);

/*
 * The per-voxel heads as one launch  (models/modules.py:273-311 Linear4xTrans: Linear(C, 4C) - LayerNorm - ReLU -
 * Linear(4C, C) - LayerNorm - ReLU - Linear(C, C_out), + the second hidden layer as a skip when C == C_out; callers
 * models/neucon_network.py:437-438 tsdf_preds / occ_preds — `heads` = 2: both run on the same rows — and :546-548
 * panoptic_preds).  A wave owns 16 voxels; the chain stays in its registers (csrc/heads.hip).
 *   x f32[n][ld_x]: the first `channels` columns of a row are the input.  y of head h: f32[n][ld_y], out_channels columns.
 *   Weights are handed over PACKED for the kernel's operand order (eprecon_amd/sparse.py pack_mlp4x; all 16-byte aligned):
 *     layer with weight Wt f32[K][M] (in x out)  ->  f32[ceil(M/16)][ceil(K/16)][64][4],
 *     block (t, c), lane l = 16 q + m, component i  =  Wt[16 c + 4 q + i][16 t + m]   (0 outside the matrix);
 *     bias / LayerNorm weight / LayerNorm bias vectors padded with zeros to a multiple of 16.
 * Shapes taken (eprecon_mlp4x_supported): channels 24 / 48 / 96 with out_channels <= 16, channels 48 / 88 / 176 with
 * out_channels 33..48; EPRECON_ERR_ARG otherwise (the caller keeps those on separate launches).
 */
typedef struct eprecon_mlp4x_head {
    const float *w1; const float *b1; const float *g1; const float *be1;   /* Linear(C, 4C), LayerNorm(4C) */
    const float *w2; const float *b2; const float *g2; const float *be2;   /* Linear(4C, C), LayerNorm(C) */
    const float *w3; const float *b3;                                      /* Linear(C, C_out) */
    float *y; int64_t ld_y;
} eprecon_mlp4x_head;
typedef struct eprecon_mlp4x_desc {
    const float *x; int64_t ld_x; int64_t n;
    int channels; int out_channels; int heads; int residual;
    float eps1; float eps2;
    eprecon_mlp4x_head head[2];
} eprecon_mlp4x_desc;
int eprecon_mlp4x_supported(int channels, int out_channels);
int eprecon_mlp4x_async(const eprecon_mlp4x_desc *desc, void *stream);

/* ------------------------------------------------------------------------------------------
 * The 2D feeder's memory-bound layers  (models/backbone.py:22-77 MnasMulti, called once per view in train mode:
 * models/neuralrecon.py:53-54, main.py:357).  Channels-last maps x f32[n][h][w][c], c % 4 == 0, 16-byte aligned.
 *
 * Replaces  nn.Conv2d(groups == channels) of torchvision's MNASNet inverted-residual blocks and the nn.BatchNorm2d behind
 *           every convolution of the trunk, for a batch of views whose statistics must not mix (one BatchNorm call per
 *           view in the reference).
 *
 * eprecon_bn2d_views_stats_async: rows [views][rows_per_view][channels] (a view's images x pixels) -> affine_out
 *   f32[views][2][channels] = (scale, shift) of the train-mode BatchNorm of each view (biased variance, eps inside the root):
 *   y = x * scale + shift.  workspace: eprecon_bn2d_views_workspace_bytes.  Two launches (range summaries, merge in range
 *   order): deterministic.  channels <= 480.
 * eprecon_bn2d_views_apply_async: out = [relu](x * scale + shift) [+ residual]; out may be x.
 * eprecon_dwconv2d_nhwc_async: depthwise ksize x ksize (3 or 5), stride 1 or 2, zero padding ksize / 2, no bias;
 *   weight_taps f32[ksize * ksize][channels] (tap-major: torch's [C,1,k,k] weight transposed).  affine (optional,
 *   f32[views][2][channels]): the producer's pending BatchNorm (+ ReLU when relu != 0) applied to every loaded input value
 *   (padding stays zero); image i belongs to view i / imgs_per_view.  out f32[n][ho][wo][channels].
 * ------------------------------------------------------------------------------------------ */
int eprecon_bn2d_views_chunks(int64_t rows_per_view, int channels);
size_t eprecon_bn2d_views_workspace_bytes(int views, int64_t rows_per_view, int channels);
int eprecon_bn2d_views_stats_async(const float *x, int views, int64_t rows_per_view, int channels, const float *gamma,
                                   const float *beta, float eps, float *affine_out, void *workspace,
                                   size_t workspace_bytes, void *stream);
int eprecon_bn2d_views_apply_async(const float *x, int views, int64_t rows_per_view, int channels, const float *affine, int relu,
                                   const float *residual, float *out, void *stream);
int eprecon_dwconv2d_nhwc_async(const float *x, int n, int height, int width, int channels, const float *weight_taps, int ksize,
                                int stride, const float *affine, int imgs_per_view, int relu, float *out, void *stream);

/*
 * The rest of a GRU-fusion level's geometry after its host read (sizes known), in one call (csrc/gru_stage_finish.hip): for the
 * two voxelisations shared by the six SConv3d of the level's ConvGRUs — CSR point lists (offsets int32[m+1], order int32[n]),
 * 3x3x3 kernel maps int32[27][m], the first voxelisation's trilinear corner tables (idx8 int32[n][8], weight8 f32[n][8]) and
 *   literal != 0 (the reference's convr, models/modules.py:216-217): the hash order of both sets (perm / rank int32[m]) and
 *                idx8_2 = the FIRST tables' indices carried to the second set in hash order (eprecon_remap_index_async);
 *   literal == 0: the second voxelisation's own corner tables (idx8_2, weight8_2 from scaled2).
 * Replaces the same sequence of single calls (eprecon_segment_lists_async, eprecon_kernel_map_async,
 * eprecon_trilinear_map_async, eprecon_sphash_order_async, eprecon_remap_index_async): bit-identical outputs.
 * The second voxelisation's chain is issued on a side stream the library owns (one per caller stream, forked from and joined to
 * `stream` with events inside the call): on return everything is ordered on `stream` as usual.  Like every call that forks, it
 * must not be issued while `stream` is being captured into a HIP graph.  (eprecon_spvcnn_geometry_async does the same.)
 */
typedef struct eprecon_gru_finish_desc {
    int64_t n; int64_t m1; int64_t m2;
    const int32_t *inverse1; const int32_t *inverse2;      /* int32[n]: point -> voxel */
    const int32_t *uniq1; const int32_t *uniq2;            /* int32[m][4] */
    const void *table1; const void *table2; uint32_t table_capacity;
    const float *scaled1; const float *scaled2;            /* f32[n][4] points in voxel units (scaled2: literal == 0 only) */
    int literal;
    int32_t *offsets1; int32_t *order1; int32_t *offsets2; int32_t *order2;
    int32_t *nbr1; int32_t *nbr2;
    int32_t *idx8_1; float *weight8_1; int32_t *idx8_2; float *weight8_2;
    int32_t *perm1; int32_t *rank1; int32_t *perm2; int32_t *rank2;   /* literal != 0 only */
    void *workspace; size_t workspace_bytes;
} eprecon_gru_finish_desc;
size_t eprecon_gru_stage_finish_workspace_bytes(int64_t n, int64_t m1, int64_t m2);
int eprecon_gru_stage_finish_async(const eprecon_gru_finish_desc *desc, void *stream);

/*
 * The BODY of one SPVCNN pass as ONE stream-ordered call (models/modules.py:148-175: stem, two k2s2 stages with residual
 * blocks :46-72, two transposed stages with skip concatenation, three scatter-means and three trilinear devoxelisations
 * ops/torchsparse_utils.py:40-105, two point MLPs models/modules.py:125-136; every BatchNorm in train mode, main.py:357).
 * Replaces ~115 calls issued one by one from the host: 27 x eprecon_conv_desc_async, eprecon_batchnorm_finalize_affine_async,
 * eprecon_batchnorm_apply_partials[_res]_async, eprecon_segment_mean_async, eprecon_devoxelize_async — the same entry points,
 * descriptors and order, issued from inside the library (bit-identical results).  The geometry comes from
 * eprecon_spvcnn_geometry_async.  conv[] / bn[] slots, in launch order:
 *    0 stem | 1 stage1 down, 2-4 res (conv, conv, 1x1 skip), 5-6 res | 7 stage2 down, 8-10 res, 11-12 res | 13 point MLP 0 |
 *   14 up1 transposed, 15-17 res, 18-19 res | 20 up2 transposed, 21-23 res, 24-25 res | 26 point MLP 1
 * weight f32[kvol][cin][cout] (kvol 27: x-fastest offsets, 8: z-fastest, 1: a [cin][cout] matrix); packed_weight (kvol 27) and
 * packed_weight16 (kvol 27 or 1, cout <= 64) are the operand-order copies of eprecon_conv_pack_weight[16]_async.
 * cs[5]: the channel plan (32, 64, 128, 96, 96) x cr.  feat f32[n, cin] point features, out f32[n, cs[4]].
 * workspace: eprecon_spvcnn_forward_workspace_bytes(desc) bytes, 256-byte aligned (every intermediate of the pass).
 */
#define EPRECON_SPVCNN_CONVS 27
typedef struct eprecon_spvcnn_conv {
    const float *weight; const float *packed_weight; const float *packed_weight16;
    int kvol; int cin; int cout;
} eprecon_spvcnn_conv;
typedef struct eprecon_spvcnn_bn {
    const float *gamma; const float *beta; float eps;
} eprecon_spvcnn_bn;
typedef struct eprecon_spvcnn_forward_desc {
    int64_t n, n1, n2, n4;
    int cin; int cs[5];
    const float *feat; int ld_feat;
    const int32_t *offsets1, *order1, *offsets4, *order4;
    const int32_t *k1, *k2, *k4, *down12, *up21, *down24, *up42;
    const int32_t *idx8_1; const float *weight8_1; const int32_t *idx8_4; const float *weight8_4;
    eprecon_spvcnn_conv conv[EPRECON_SPVCNN_CONVS];
    eprecon_spvcnn_bn bn[EPRECON_SPVCNN_CONVS];
    float *out; int ld_out;
    void *workspace; size_t workspace_bytes;
} eprecon_spvcnn_forward_desc;
size_t eprecon_spvcnn_forward_workspace_bytes(const eprecon_spvcnn_forward_desc *desc);
int eprecon_spvcnn_forward_async(const eprecon_spvcnn_forward_desc *desc, void *stream);

/*
 * The geometry of one SPVCNN pass after its host read (the sizes n1, n2, n4 of the voxel sets at tensor strides 1, 2, 4, numbered
 * by three queued eprecon_unique_coords(_dn)_async calls), in one call (csrc/spvcnn_geometry.hip): CSR point lists of strides 1
 * and 4 (offsets int32[m+1], order int32[n]; idx4 int32[n] = the stride-4 voxel of every point), the k2s2 maps down12 int32[8][n2],
 * down24 int32[8][n4] and their transposes up21 int32[8][n1], up42 int32[8][n2], the 3x3x3 kernel maps k1 / k2 / k4
 * int32[27][n1 / n2 / n4], trilinear corner tables of strides 1 and 4 (int32[n][8], f32[n][8]).  parent2 int32[n1] / parent4
 * int32[n2]: the inverse index of the stride-2 / stride-4 numbering.  Replaces the same sequence of single calls: bit-identical.
 */
typedef struct eprecon_spvcnn_geometry_desc {
    int64_t n; int64_t n1; int64_t n2; int64_t n4;
    const float *scaled; const int32_t *vox; const int32_t *inverse1;     /* points: f32[n][4], int32[n][4], int32[n] */
    const int32_t *coords1; const int32_t *coords2; const int32_t *coords4;
    const int32_t *parent2; const int32_t *parent4;
    const void *table1; const void *table2; const void *table4;
    uint32_t capacity1; uint32_t capacity2; uint32_t capacity4;
    int32_t *offsets1; int32_t *order1; int32_t *idx4; int32_t *offsets4; int32_t *order4;
    int32_t *down12; int32_t *up21; int32_t *down24; int32_t *up42;
    int32_t *k1; int32_t *k2; int32_t *k4;
    int32_t *idx8_1; float *weight8_1; int32_t *idx8_4; float *weight8_4;
    void *workspace; size_t workspace_bytes;
} eprecon_spvcnn_geometry_desc;
size_t eprecon_spvcnn_geometry_workspace_bytes(int64_t n, int64_t n1, int64_t n4);
int eprecon_spvcnn_geometry_async(const eprecon_spvcnn_geometry_desc *desc, void *stream);

/* x2 bilinear upsampling of channels-last maps, in f32[n,h,w,c] -> out f32[n,2h,2w,c], c % 4 == 0
 * (F.interpolate(scale_factor=2, mode="bilinear") in feat_fusion_pre,
 * models/occupancy_initialization.py:46) */
int eprecon_upsample2x_nhwc_async(const float *in, float *out, int n, int h, int w, int channels,
                                  void *stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the sparse operators  (training, main.py:181-313 loss.backward())
 *
 * Replaces  the autograd of spnn.Conv3d / spconv SubMConv3d, torchsparse.nn.functional.spdevoxelize and
 *           spvoxelize (ops/torchsparse_utils.py:35,63,100), which the reference gets from the two libraries.
 *
 * For y[i] = bias + sum_k x[nbr[k][i]] @ W[k]:
 *   dW[k]  = sum_i x[nbr[k][i]]^T dy[i]                eprecon_sparse_conv_wgrad_async (fp32 MFMA, deterministic:
 *                                                       row chunks reduced in chunk order through `workspace`)
 *   dx[j]  = sum_k dy[inv[k][j]] @ W[k]^T              eprecon_sparse_conv_async on the inverted map
 *   inv[k][j] = i  <=>  nbr[k][i] = j                  eprecon_invert_map_async (a kernel map is injective per offset)
 *   dbias  = column sums of dy                          (host: one reduction)
 * devoxelize:  d voxel_feat[idx8[p][c]] += weight8[p][c] * d out[p]   (hardware float atomics; dvoxel_feat is zeroed)
 * segment mean (voxelize): d point_feat[p] = d mean[idx[p]] * scale[idx[p]], scale = 1 / count
 * ------------------------------------------------------------------------------------------ */
size_t eprecon_sparse_conv_wgrad_workspace_bytes(int kvol, int64_t n_out, int cin, int cout);
int eprecon_sparse_conv_wgrad_async(const float *x, int ld_x, const float *dy, int ld_dy, const int32_t *nbr, int kvol,
                                    int64_t n_out, int cin, int cout, float *dweight, void *workspace,
                                    size_t workspace_bytes, void *stream);
int eprecon_invert_map_async(const int32_t *nbr, int kvol, int64_t n_out, int64_t n_in, int32_t *inv, void *stream);
int eprecon_devoxelize_backward_async(const float *dout, int ld_out, const int32_t *idx8, const float *weight8, int64_t n,
                                      int channels, int64_t n_voxels, float *dvoxel_feat, int ld_feat, void *stream);
/* the same without float atomics (run-to-run bit-identical): offsets int32[n_voxels + 1] / order int32[<= 8 n] = the CSR lists
 * of eprecon_segment_lists_async over the flattened idx8 (entry e = 8 p + corner; entries of missing corners are skipped);
 * dvoxel_feat[v] = sum over the voxel's entries in list order of weight8[e] * dout[e / 8].  The lists depend on the corner
 * table only: build once, reuse for every layer that devoxelises with it. */
int eprecon_devoxelize_backward_csr_async(const float *dout, int ld_out, const float *weight8, const int32_t *offsets,
                                          const int32_t *order, int64_t n_voxels, int channels, float *dvoxel_feat, int ld_feat,
                                          void *stream);
int eprecon_gather_rows_scaled_async(const float *src, int ld_src, const int32_t *idx, const float *scale, int64_t n,
                                     int channels, float *dst, int ld_dst, void *stream);
/* back-projection: d feats[view][b][y][x][c] (channels-last, zeroed here) from d out f32[n_valid, ld_dout] (and, mode
 * VARIANCE, the optional d mean f32[n_valid, C]) at the VALID voxels `coords_valid` the forward returned
 * (ops/back_project.py:47-61 / models/occupancy_initialization.py:113-128 under autograd: grid_sample's backward,
 * the masked mean / variance).  Hardware float atomics. */
int eprecon_back_project_backward_async(const int32_t *coords_valid, int64_t n_valid, const float *origin, int batch,
                                        float voxel_size, const float *feats_nhwc, const float *krcam, int n_views,
                                        int channels, int height, int width, int mode, const float *dout, int ld_dout,
                                        const float *dmean, float *dfeats_nhwc, void *stream);
/* the same, run-to-run bit-identical: contributions are accumulated as 64-bit fixed point (2^-40 resolution, |sum| < 8.4e6) with
 * integer atomics — order-independent — in `workspace` (eprecon_back_project_backward_workspace_bytes) and converted at the end */
size_t eprecon_back_project_backward_workspace_bytes(int batch, int n_views, int channels, int height, int width);
int eprecon_back_project_backward_det_async(const int32_t *coords_valid, int64_t n_valid, const float *origin, int batch,
                                            float voxel_size, const float *feats_nhwc, const float *krcam, int n_views,
                                            int channels, int height, int width, int mode, const float *dout, int ld_dout,
                                            const float *dmean, float *dfeats_nhwc, void *workspace, size_t workspace_bytes,
                                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EPRECON_HIP_H */
