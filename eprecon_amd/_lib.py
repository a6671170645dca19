"""ctypes binding of libeprecon_hip.so (the C ABI in include/eprecon_hip.h).

There is NO fallback: if the library is missing or was not built for gfx950 the import of any
operator raises.  torch is imported first so that the library binds to the same libamdhip64.so.7
(the one bundled with PyTorch-ROCm) instead of bringing a second HIP runtime into the process.
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# EPRECON_LIB_PATH: an A/B twin of the library built by `python -m eprecon_amd.build --variant NAME [-DX ...]` under
# gpurun_out/variants/ (timing tools only; nothing but libeprecon_hip.so sits beside the package)
LIB_PATH = os.environ.get("EPRECON_LIB_PATH") or os.path.join(_HERE, "libeprecon_hip.so")

_c = ctypes
_vp, _i, _i64, _f, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); must list every symbol include/eprecon_hip.h declares
# (tests/test_cabi_symbols.py checks the two against each other)
SIGNATURES = {
    "eprecon_abi_version": (_i, []),
    "eprecon_build_arch": (_c.c_char_p, []),
    "eprecon_back_project_workspace_bytes": (_sz, [_i64, _i, _i, _i, _i, _i, _i]),
    "eprecon_back_project_async": (_i, [_vp, _i64, _vp, _i, _f, _vp, _i, _vp, _i, _i, _i, _i, _i, _i,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "eprecon_back_project": (_i, [_vp, _i64, _vp, _i, _f, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "eprecon_profile_gather_kernel": (_c.c_char_p, []),
    "eprecon_hash_capacity": (_c.c_uint32, [_i64]),
    "eprecon_hash_table_bytes": (_sz, [_c.c_uint32]),
    "eprecon_hash_build_async": (_i, [_vp, _i64, _i, _vp, _c.c_uint32, _vp]),
    "eprecon_hash_build_dn_async": (_i, [_vp, _i64, _vp, _i, _vp, _c.c_uint32, _vp]),
    "eprecon_hash_query_async": (_i, [_vp, _c.c_uint32, _vp, _i64, _i, _vp, _vp]),
    "eprecon_hash_status": (_i, [_vp, _vp]),
    "eprecon_unique_workspace_bytes": (_sz, [_i64]),
    "eprecon_unique_coords_async": (_i, [_vp, _i64, _i, _vp, _c.c_uint32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "eprecon_unique_coords_dn_async": (_i, [_vp, _i64, _vp, _i, _vp, _c.c_uint32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "eprecon_unique_hierarchy_dn_async": (_i, [_vp, _i64, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "eprecon_point_quantize_dn_async": (_i, [_vp, _i64, _vp, _f, _vp, _vp, _vp]),
    "eprecon_spvcnn_points_dn_async": (_i, [_vp, _i64, _vp, _i, _i, _vp, _i, _f, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eprecon_gru_stage_capacity": (_i64, [_vp, _i64, _i]),
    "eprecon_gru_stage_workspace_bytes": (_sz, [_i64]),
    "eprecon_gru_stage_begin_async": (_i, [_vp, _vp]),
    "eprecon_gru_stage_commit_async": (_i, [_vp, _vp, _vp, _vp]),
    "eprecon_kernel_map_async": (_i, [_vp, _c.c_uint32, _vp, _i64, _i, _i, _vp, _vp]),
    "eprecon_kernel_map_self_async": (_i, [_vp, _c.c_uint32, _vp, _i64, _i, _vp, _vp]),
    "eprecon_transpose_map_async": (_i, [_vp, _i64, _vp, _i, _vp, _vp]),
    "eprecon_sparse_conv_async": (_i, [_vp, _i64, _i, _vp, _i, _i64, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "eprecon_conv_bn_partial_bytes": (_sz, [_i64, _i]),
    "eprecon_sparse_conv_fused_async": (_i, [_vp, _i64, _i, _vp, _i, _i64, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i,
                                             _vp, _vp]),
    "eprecon_conv_desc_async": (_i, [_vp, _vp]),
    "eprecon_exclusive_scan_async": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "eprecon_conv_desc_partial_rows": (_i64, [_vp]),
    "eprecon_bn_acc_words": (_sz, [_i]),
    "eprecon_conv_desc_takes_bn_acc": (_i, [_vp]),
    "eprecon_batchnorm_acc_affine_async": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "eprecon_affine_rows_acc_async": (_i, [_vp, _i64, _i, _i, _vp, _i, _i, _f, _i, _vp, _i, _vp]),
    "eprecon_affine_rows_res_async": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "eprecon_batchnorm_finalize_affine_async": (_i, [_vp, _i64, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "eprecon_affine_rows_async": (_i, [_vp, _i64, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "eprecon_pixel_map_async": (_i, [_i, _i, _i, _i, _vp, _vp]),
    "eprecon_batchnorm_apply_workspace_bytes": (_sz, [_i]),
    "eprecon_batchnorm_apply_partials_async": (_i, [_vp, _i64, _i, _i, _vp, _i64, _vp, _vp, _f, _vp, _i, _i, _vp, _i,
                                                    _vp, _vp, _vp, _sz, _vp]),
    "eprecon_batchnorm_apply_partials_res_async": (_i, [_vp, _i64, _i, _i, _vp, _i64, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _vp,
                                                        _i, _vp, _sz, _vp]),
    "eprecon_batchnorm_workspace_bytes": (_sz, [_i64, _i]),
    "eprecon_batchnorm_train_async": (_i, [_vp, _i64, _i, _i, _vp, _vp, _f, _vp, _i, _i, _vp, _i, _vp, _vp,
                                           _vp, _sz, _vp]),
    "eprecon_rowwise_layernorm_async": (_i, [_vp, _i64, _i, _i, _vp, _i, _vp, _vp, _f, _i, _i, _vp, _i, _vp]),
    "eprecon_init_select_workspace_bytes": (_sz, [_i, _i]),
    "eprecon_init_select_async": (_i, [_vp, _vp, _i64, _f, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "eprecon_upsample_async": (_i, [_vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "eprecon_aligned_coords_async": (_i, [_vp, _i64, _vp, _i, _f, _vp, _vp, _vp]),
    "eprecon_point_quantize_async": (_i, [_vp, _i64, _f, _vp, _vp, _vp]),
    "eprecon_segment_workspace_bytes": (_sz, [_i64, _i64]),
    "eprecon_segment_lists_async": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "eprecon_segment_mean_async": (_i, [_vp, _i, _vp, _vp, _i64, _i, _vp, _i, _vp]),
    "eprecon_sphash_async": (_i, [_vp, _i64, _vp, _vp]),
    "eprecon_sphash_order_workspace_bytes": (_sz, [_i64]),
    "eprecon_sphash_order_async": (_i, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "eprecon_remap_index_async": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "eprecon_trilinear_map_async": (_i, [_vp, _c.c_uint32, _vp, _i64, _i, _vp, _vp, _vp]),
    "eprecon_devoxelize_async": (_i, [_vp, _i, _vp, _vp, _i64, _i, _vp, _i, _i, _vp]),
    "eprecon_sparse_conv_wgrad_workspace_bytes": (_sz, [_i, _i64, _i, _i]),
    "eprecon_sparse_conv_wgrad_async": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _i, _vp, _vp, _sz, _vp]),
    "eprecon_invert_map_async": (_i, [_vp, _i, _i64, _i64, _vp, _vp]),
    "eprecon_devoxelize_backward_async": (_i, [_vp, _i, _vp, _vp, _i64, _i, _i64, _vp, _i, _vp]),
    "eprecon_gather_rows_scaled_async": (_i, [_vp, _i, _vp, _vp, _i64, _i, _vp, _i, _vp]),
    "eprecon_back_project_backward_async": (_i, [_vp, _i64, _vp, _i, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "eprecon_back_project_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "eprecon_back_project_backward_det_async": (_i, [_vp, _i64, _vp, _i, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _sz,
                                                     _vp]),
    "eprecon_devoxelize_backward_csr_async": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _i, _vp, _i, _vp]),
    "eprecon_devoxelize_gate_async": (_i, [_vp, _i, _vp, _vp, _i64, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "eprecon_devoxelize_gate_tail_async": (_i, [_vp, _i, _vp, _vp, _i64, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i,
                                                _i, _vp]),
    "eprecon_fbv_union_workspace_bytes": (_sz, [_i]),
    "eprecon_fbv_union_async": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _sz, _vp]),
    "eprecon_map_set_fragment": (_i, [_vp, _i]),
    "eprecon_map_stamps_async": (_i, [_vp, _vp, _vp, _i, _c.c_int32, _vp]),
    "eprecon_map_select_boundary_async": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "eprecon_map_pack_boundary_async": (_i, [_vp, _vp, _i64, _vp]),
    "eprecon_map_merge_boundary": (_i, [_vp, _vp, _i64, _vp, _i, _vp, _vp]),
    "eprecon_map_create": (_i, [_i, _c.POINTER(_vp)]),
    "eprecon_map_destroy": (_i, [_vp]),
    "eprecon_map_reset": (_i, [_vp]),
    "eprecon_map_size": (_i64, [_vp]),
    "eprecon_map_channels": (_i, [_vp]),
    "eprecon_map_export_async": (_i, [_vp, _vp, _vp, _vp]),
    "eprecon_map_import_async": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "eprecon_map_crop_union": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eprecon_map_gather_async": (_i, [_vp, _vp, _i64, _i, _i, _f, _vp, _i, _vp]),
    "eprecon_map_update_async": (_i, [_vp, _vp, _i64, _vp, _i, _vp]),
    "eprecon_map_target_fuse": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i64, _vp, _vp]),
    "eprecon_gather_rows_async": (_i, [_vp, _i, _vp, _i64, _i, _f, _vp, _i, _vp]),
    "eprecon_tsdf_integrate_async": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp, _vp, _f, _f, _i, _vp, _vp]),
    "eprecon_marching_cubes_table": (_i, [_vp]),
    "eprecon_marching_cubes_workspace_bytes": (_sz, [_i, _i, _i]),
    "eprecon_marching_cubes_count": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "eprecon_marching_cubes_emit_async": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eprecon_nearest_voxel_async": (_i, [_vp, _c.c_uint32, _vp, _i64, _vp, _i64, _i, _vp, _vp]),
    "eprecon_upsample2x_nhwc_async": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "eprecon_decoder_keys_async": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "eprecon_decoder_query_side_async": (_i, [_vp, _vp]),
    "eprecon_panoptic_stats_async": (_i, [_vp, _i64, _vp, _vp, _i, _i64, _vp, _vp, _vp, _vp]),
    "eprecon_panoptic_assign_async": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "eprecon_masked_attention_workspace_bytes": (_sz, [_i64, _i, _i, _i]),
    "eprecon_masked_attention_async": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i64, _vp, _i, _vp, _i64, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "eprecon_profile_enable": (_i, [_i]),
    "eprecon_profile_gather_ms": (_f, []),
    "eprecon_profile_conv_arm": (_i, [_i, _i, _i, _i64]),
    "eprecon_profile_conv_ms": (_f, [_c.POINTER(_i64), _c.POINTER(_c.c_char_p)]),
    "eprecon_profile_conv_pairs": (_i64, []),
    "eprecon_mlp4x_supported": (_i, [_i, _i]),
    "eprecon_spvcnn_forward_workspace_bytes": (_sz, [_vp]),
    "eprecon_spvcnn_forward_async": (_i, [_vp, _vp]),
    "eprecon_spvcnn_geometry_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "eprecon_spvcnn_geometry_async": (_i, [_vp, _vp]),
    "eprecon_gru_stage_finish_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "eprecon_gru_stage_finish_async": (_i, [_vp, _vp]),
    "eprecon_bn2d_views_chunks": (_i, [_i64, _i]),
    "eprecon_bn2d_views_workspace_bytes": (_sz, [_i, _i64, _i]),
    "eprecon_bn2d_views_stats_async": (_i, [_vp, _i, _i64, _i, _vp, _vp, _f, _vp, _vp, _sz, _vp]),
    "eprecon_bn2d_views_apply_async": (_i, [_vp, _i, _i64, _i, _vp, _i, _vp, _vp, _vp]),
    "eprecon_dwconv2d_nhwc_async": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "eprecon_mlp4x_async": (_i, [_vp, _vp]),
    "eprecon_profile_conv_executed_pairs": (_i64, []),
    "eprecon_profile_last_conv_kernel": (_c.c_char_p, []),
    "eprecon_profile_mark_async": (_i, [_i, _vp]),
    "eprecon_conv_desc_workspace_bytes": (_sz, [_vp]),
    "eprecon_sparsify_workspace_bytes": (_sz, [_i64]),
    "eprecon_sparsify_async": (_i, [_vp, _i, _f, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _sz, _vp]),
    "eprecon_grid_rank_async": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "eprecon_conv_pack_weight_floats": (_sz, [_i, _i, _i]),
    "eprecon_conv_pack_weight_async": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "eprecon_conv_pack_weight16_floats": (_sz, [_i, _i, _i]),
    "eprecon_conv_pack_weight16_async": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "eprecon_conv_pack_many_async": (_i, [_vp, _i, _vp]),
    "eprecon_nchw_to_nhwc_async": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "eprecon_views_to_rows_async": (_i, [_vp, _vp]),
}

EPRECON_OK, EPRECON_EMPTY = 0, 1
ABI_VERSION = 1


class EpreconError(RuntimeError):
    pass


_lib = None


def load():
    """Returns the loaded library; raises EpreconError when it is absent (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EpreconError(
            f"{LIB_PATH} not found: build it with `python -m eprecon_amd.build` "
            "(hipcc --offload-arch=gfx950). eprecon_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.eprecon_abi_version() != ABI_VERSION:
        raise EpreconError("libeprecon_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    """0 -> True, EPRECON_EMPTY -> False (caller returns None like the reference), <0 raises."""
    if rc == EPRECON_OK:
        return True
    if rc == EPRECON_EMPTY:
        return False
    if rc <= -1000:
        raise EpreconError(f"{what}: HIP error {-(rc + 1000)}")
    raise EpreconError(f"{what}: error {rc} "
                       "(-1 bad argument, -2 workspace too small, -3 unsupported size)")


def ptr(t):
    """device pointer of a torch tensor (or None) as the plain int ctypes converts to void*"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def current_stream():
    """hipStream_t of torch's current stream on the current device, as an int.  The raw getter is used when this
    torch build has it: torch.cuda.current_stream() builds a Stream object and costs ~8 us per call, which at
    ~5,000 launches per fragment was 3.7 ms of host time per fragment (tools/hostprof_cfg4.py)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device()) or None
    return torch.cuda.current_stream().cuda_stream or None


class ViewsDesc(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_views_desc"""
    _fields_ = [("src", (ctypes.c_void_p * 16) * 3), ("dst", ctypes.c_void_p * 3), ("channels", ctypes.c_int32 * 3),
                ("hw", ctypes.c_int32 * 3), ("levels", ctypes.c_int32), ("n_views", ctypes.c_int32)]


class ConvDesc(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_conv_desc"""
    _fields_ = [("x", ctypes.c_void_p), ("n_in", ctypes.c_int64), ("ld_x", ctypes.c_int),
                ("nbr", ctypes.c_void_p), ("kvol", ctypes.c_int), ("n_out", ctypes.c_int64),
                ("weight", ctypes.c_void_p), ("cin", ctypes.c_int), ("cout", ctypes.c_int),
                ("bias", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int),
                ("out", ctypes.c_void_p), ("ld_out", ctypes.c_int),
                ("relu", ctypes.c_int), ("accumulate", ctypes.c_int),
                ("in_scale", ctypes.c_void_p), ("in_shift", ctypes.c_void_p), ("in_relu", ctypes.c_int),
                ("res_scale", ctypes.c_void_p), ("res_shift", ctypes.c_void_p), ("res_relu", ctypes.c_int),
                ("bn_partial", ctypes.c_void_p),
                ("ln", ctypes.c_int), ("ln_gamma", ctypes.c_void_p), ("ln_beta", ctypes.c_void_p),
                ("ln_eps", ctypes.c_float), ("ln_post_relu", ctypes.c_int),
                ("img_h", ctypes.c_int), ("img_w", ctypes.c_int), ("img_maps", ctypes.c_int),
                ("vox_rank", ctypes.c_void_p), ("grid_x", ctypes.c_int), ("grid_y", ctypes.c_int), ("grid_z", ctypes.c_int),
                ("packed_weight", ctypes.c_void_p), ("packed_weight16", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("bn_acc", ctypes.c_void_p), ("bn_acc_ld", ctypes.c_int), ("bn_acc_c0", ctypes.c_int),
                ("bn_gamma", ctypes.c_void_p), ("bn_beta", ctypes.c_void_p),
                ("in_acc", ctypes.c_void_p), ("in_acc_ld", ctypes.c_int), ("in_acc_c0", ctypes.c_int),
                ("in_eps", ctypes.c_float), ("in_affine_scratch", ctypes.c_void_p)]


class GruStageDesc(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_gru_stage_desc"""
    _fields_ = [("map", ctypes.c_void_p), ("target_map", ctypes.c_void_p),
                ("cur_coords", ctypes.c_void_p), ("cur_feat", ctypes.c_void_p), ("n_cur", ctypes.c_int64), ("ld_cur", ctypes.c_int),
                ("dim", ctypes.c_int), ("interval", ctypes.c_int), ("activity_mode", ctypes.c_int),
                ("rel", ctypes.c_int32 * 3),
                ("tsdf_gt", ctypes.c_void_p), ("occ_gt", ctypes.c_void_p),
                ("origin", ctypes.c_void_p), ("w2ac", ctypes.c_void_p),
                ("voxel_size", ctypes.c_float), ("resolution", ctypes.c_float),
                ("ch_voxel", ctypes.c_int), ("batch_index", ctypes.c_int),
                ("capacity", ctypes.c_int64),
                ("updated", ctypes.c_void_p), ("out_coords", ctypes.c_void_p), ("r_coords", ctypes.c_void_p),
                ("hx_voxel", ctypes.c_void_p), ("hx_image", ctypes.c_void_p), ("tsdf_target", ctypes.c_void_p),
                ("scaled1", ctypes.c_void_p), ("vox1", ctypes.c_void_p), ("inverse1", ctypes.c_void_p), ("uniq1", ctypes.c_void_p),
                ("table1", ctypes.c_void_p),
                ("scaled2", ctypes.c_void_p), ("vox2", ctypes.c_void_p), ("inverse2", ctypes.c_void_p), ("uniq2", ctypes.c_void_p),
                ("table2", ctypes.c_void_p),
                ("table_capacity", ctypes.c_uint32),
                ("counts", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t)]


class DecoderLayerDesc(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_decoder_layer_desc"""
    _PTRS_A = ["o_attn", "state_in", "query_pos", "cross_out_wt", "cross_out_b", "cross_ln_g", "cross_ln_b", "self_in_wt", "self_in_b",
               "self_out_wt", "self_out_b", "self_ln_g", "self_ln_b", "ffn1_wt", "ffn1_b", "ffn2_wt", "ffn2_b", "ffn_ln_g", "ffn_ln_b",
               "dec_ln_g", "dec_ln_b", "cls_wt", "cls_b", "m1_wt", "m1_b", "m2_wt", "m2_b", "m3_wt", "m3_b", "next_q_wt", "next_q_b"]
    _PTRS_B = ["state_out", "cls_out", "mask_embed_out", "next_q_out", "workspace"]
    _fields_ = ([(n, ctypes.c_int) for n in ("n_queries", "channels", "n_heads", "ffn_dim", "n_class_logits", "mask_hidden")]
                + [(n, ctypes.c_void_p) for n in _PTRS_A] + [("ln_eps", ctypes.c_float)] + [(n, ctypes.c_void_p) for n in _PTRS_B])


class GruFinishDesc(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_gru_finish_desc"""
    _fields_ = ([(n, ctypes.c_int64) for n in ("n", "m1", "m2")]
                + [(n, ctypes.c_void_p) for n in ("inverse1", "inverse2", "uniq1", "uniq2", "table1", "table2")]
                + [("table_capacity", ctypes.c_uint32)] + [(n, ctypes.c_void_p) for n in ("scaled1", "scaled2")]
                + [("literal", ctypes.c_int)]
                + [(n, ctypes.c_void_p) for n in ("offsets1", "order1", "offsets2", "order2", "nbr1", "nbr2", "idx8_1", "weight8_1",
                                                  "idx8_2", "weight8_2", "perm1", "rank1", "perm2", "rank2", "workspace")]
                + [("workspace_bytes", ctypes.c_size_t)])


class SpvcnnGeometryDesc(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_spvcnn_geometry_desc"""
    _fields_ = ([(n, ctypes.c_int64) for n in ("n", "n1", "n2", "n4")]
                + [(n, ctypes.c_void_p) for n in ("scaled", "vox", "inverse1", "coords1", "coords2", "coords4", "parent2", "parent4",
                                                  "table1", "table2", "table4")]
                + [(n, ctypes.c_uint32) for n in ("capacity1", "capacity2", "capacity4")]
                + [(n, ctypes.c_void_p) for n in ("offsets1", "order1", "idx4", "offsets4", "order4", "down12", "up21", "down24", "up42",
                                                  "k1", "k2", "k4", "idx8_1", "weight8_1", "idx8_4", "weight8_4", "workspace")]
                + [("workspace_bytes", ctypes.c_size_t)])


SPVCNN_CONVS = 27


class SpvcnnConv(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_spvcnn_conv"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("weight", "packed_weight", "packed_weight16")] + [(n, ctypes.c_int) for n in ("kvol", "cin", "cout")]


class SpvcnnBn(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_spvcnn_bn"""
    _fields_ = [("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("eps", ctypes.c_float)]


class SpvcnnForwardDesc(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_spvcnn_forward_desc"""
    _fields_ = ([(n, ctypes.c_int64) for n in ("n", "n1", "n2", "n4")] + [("cin", ctypes.c_int), ("cs", ctypes.c_int * 5)]
                + [("feat", ctypes.c_void_p), ("ld_feat", ctypes.c_int)]
                + [(n, ctypes.c_void_p) for n in ("offsets1", "order1", "offsets4", "order4", "k1", "k2", "k4", "down12", "up21", "down24",
                                                  "up42", "idx8_1", "weight8_1", "idx8_4", "weight8_4")]
                + [("conv", SpvcnnConv * SPVCNN_CONVS), ("bn", SpvcnnBn * SPVCNN_CONVS)]
                + [("out", ctypes.c_void_p), ("ld_out", ctypes.c_int), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t)])


class Mlp4xHead(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_mlp4x_head"""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "w3", "b3", "y")]
                + [("ld_y", ctypes.c_int64)])


class Mlp4xDesc(ctypes.Structure):
    """include/eprecon_hip.h: eprecon_mlp4x_desc"""
    _fields_ = [("x", ctypes.c_void_p), ("ld_x", ctypes.c_int64), ("n", ctypes.c_int64), ("channels", ctypes.c_int),
                ("out_channels", ctypes.c_int), ("heads", ctypes.c_int), ("residual", ctypes.c_int), ("eps1", ctypes.c_float),
                ("eps2", ctypes.c_float), ("head", Mlp4xHead * 2)]


# Side streams are shared by role across every model object of the process: HIP multiplexes streams onto a few hardware
# queues (four by default), and a process that builds several networks, each creating its own side streams, piles up
# fifteen of them.
SIDE_PAIR, SIDE_BRANCH_A, SIDE_BRANCH_B, SIDE_PANOPTIC, SIDE_EXCHANGE, SIDE_SETUP = range(6)
_SIDE_STREAMS = {}


def side_stream(device, role):
    """the process-wide side stream of `role` on `device`: SIDE_PAIR the second of two twin networks (ConvGRU cells, 2D
    backbones), SIDE_BRANCH_A / _B the outer levels of the 2D fusion stack, SIDE_PANOPTIC the pipelined panoptic branch,
    SIDE_EXCHANGE the boundary exchange, SIDE_SETUP one-off warm-up / graph-capture work"""
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, role)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", index))
    return st


_WORKSPACES = {}


def workspace(nbytes, device):
    """grow-only scratch buffer per (device, current stream): operators issued on different streams
    (the per-level branches of the 2D stack, the pipelined back-projections) never share scratch"""
    key = (device.type, device.index, current_stream())
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _WORKSPACES[key] = buf
    return buf


# Device-side values whose check rides on a LATER blocking read instead of costing one of their own: (int32 device element,
# expected value, message, stream the producing launch was queued on).  EVERY blocking count read of the package goes through
# read_counts(), which takes the entries queued on ITS stream along in the same transfer and raises EpreconError on a
# mismatch (entries of another stream — the pipelined panoptic worker — are left for that stream's reads: a read on stream A
# does not wait for stream B).  Guarded by a lock: the worker thread reads too.
_DEFERRED = []
_DEFERRED_LOCK = __import__("threading").Lock()
_DEFERRED_MAX = 64


def defer_check(dev_value, expected, what):
    """dev_value: a one-element device tensor (a VIEW is kept as it is: cloning it would cost a launch per check on the hot
    path; what it keeps alive until the next read of its stream — at most _DEFERRED_MAX entries — is a count buffer or a
    DenseMap rank volume of a few MB that its VoxelSet caches anyway)"""
    item = (dev_value, int(expected), what, current_stream())
    evicted = None
    with _DEFERRED_LOCK:
        _DEFERRED.append(item)
        if len(_DEFERRED) > _DEFERRED_MAX:   # a path that never reaches a read on its stream: the OLDEST entry is verified now
            evicted = _DEFERRED.pop(0)       # (one blocking read, counted) instead of being dropped unseen
    if evicted is not None:
        count_host_read()
        verify_deferred([evicted], evicted[0].to(torch.int32).tolist())


def requeue_deferred(items):
    """entries a reader took along and never verified (an abandoned PinnedRead) go back to the front of the list"""
    if items:
        with _DEFERRED_LOCK:
            _DEFERRED[:0] = items


def take_deferred(stream="current"):
    """the pending checks queued on `stream` (default: the current one; None: all of them), removed from the list"""
    st = current_stream() if stream == "current" else stream
    with _DEFERRED_LOCK:
        out = [it for it in _DEFERRED if stream is None or it[3] == st]
        _DEFERRED[:] = [it for it in _DEFERRED if not (stream is None or it[3] == st)]
    return out


def verify_deferred(items, host_values):
    for (_, expected, what, *_rest), got in zip(items, host_values):
        if int(got) != expected:
            raise EpreconError(f"{what}: expected {expected}, the device reports {int(got)} (EPRECON_ERR_ARG)")


# (Round 5 measured the reads through an asynchronous copy into pinned memory + a busy-wait on the event behind it instead of
# Tensor.tolist(): 13.79-13.82 ms per cfg4 fragment against 13.53 on the same box (gpurun r05c) — the runtime's synchronous
# hipMemcpy returns sooner than torch's event wait; not kept.)
def read_counts(counts):
    """THE blocking device -> host read of the package: `counts` (an int32 device tensor) as a flat Python list.  The pending
    deferred checks of the current stream ride along in the same transfer and are verified before the counts are returned."""
    count_host_read()
    pending = take_deferred()
    flat = counts.reshape(-1)
    if pending:
        host = torch.cat([flat] + [t.reshape(1).to(flat.dtype) for t, *_ in pending]).tolist()
        verify_deferred(pending, host[flat.numel():])
        return host[:flat.numel()]
    return flat.tolist()


class PinnedRead:
    """read_counts() without blocking now: the counts AND the deferred checks pending on the current stream at this point are
    copied to pinned host memory behind everything queued so far; result() waits for that copy only (an event), verifies the
    checks and returns the counts.  Work queued after the constructor is not waited for."""

    def __init__(self, counts):
        self._pending = take_deferred()
        flat = counts.reshape(-1)
        self._n = flat.numel()
        src = flat if not self._pending else torch.cat([flat] + [t.reshape(1).to(flat.dtype) for t, *_ in self._pending])
        self._pinned = torch.empty(src.numel(), dtype=src.dtype, pin_memory=True)
        self._pinned.copy_(src, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()

    def result(self):
        count_host_read()
        self._event.synchronize()
        host = self._pinned.tolist()
        if self._pending:
            pending, self._pending = self._pending, []
            verify_deferred(pending, host[self._n:])
        return host[:self._n]

    def __del__(self):
        # result() never called: the checks this read took along are handed back, so that the next read of their stream (or
        # drain_deferred) still verifies them
        try:
            requeue_deferred(self._pending)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def drain_deferred():
    """blocking: verify whatever is pending on the current stream (end of a step that finished without a count read)"""
    pending = take_deferred()
    if pending:
        count_host_read()
        verify_deferred(pending, torch.cat([t.reshape(1).to(torch.int32) for t, *_ in pending]).tolist())


HOST_READS = 0    # blocking device -> host reads issued by the package since import (bench.py: blocking_reads_per_fragment)


def count_host_read(n=1):
    global HOST_READS
    HOST_READS += n


def last_conv_kernel():
    """name of the kernel family the most recent convolution launch went to (eprecon_profile_last_conv_kernel)"""
    return load().eprecon_profile_last_conv_kernel().decode()
