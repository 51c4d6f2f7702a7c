"""ctypes binding of libwast3d_hip.so (include/wast3d_hip.h).  The product path has NO fallback:
if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("W3D_LIB_PATH") or os.path.join(_HERE, "libwast3d_hip.so")   # the override is a dev hook for A/B kernel timing

F32, BF16, F16, FP8 = 0, 1, 2, 3
_lib = None

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
f64 = C.c_double


class AdamGroup(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("numel", i64), ("lr", C.c_double), ("grad2", vp)]


class RasterIn(C.Structure):
    _fields_ = [("P", i32), ("sh_degree", i32), ("sh_coeffs", i32), ("H", i32), ("W", i32),
                ("background", vp), ("means3D", vp), ("shs", vp), ("colors_precomp", vp), ("opacities", vp),
                ("scales", vp), ("rotations", vp), ("cov3D_precomp", vp), ("viewmatrix", vp), ("projmatrix", vp),
                ("campos", vp), ("sampling_offsets", vp), ("scale_modifier", f32), ("tan_fovx", f32), ("tan_fovy", f32),
                ("prefiltered", i32), ("shs_rest", vp)]


class RasterGradIn(C.Structure):
    _fields_ = [("dL_dout_color", vp), ("dL_dout_depth", vp)]


class RasterGradOut(C.Structure):
    _fields_ = [("dL_dmeans2D", vp), ("dL_dcolors", vp), ("dL_dopacity", vp), ("dL_dmeans3D", vp), ("dL_dcov3D", vp),
                ("dL_dsh", vp), ("dL_dscales", vp), ("dL_drotations", vp), ("accum", vp), ("dL_dsh_rest", vp)]


ALLOC_FN = C.CFUNCTYPE(vp, vp, sz)

# name -> (restype, argtypes); must list every symbol include/wast3d_hip.h declares (tests check this)
SIGNATURES = {
    "w3d_strerror": (C.c_char_p, [i32]),
    "w3d_abi_version": (i32, []),
    "w3d_last_hip_error_string": (C.c_char_p, []),
    "w3d_clear_hip_error": (None, []),
    "w3d_knn_workspace_bytes": (sz, [i64, i64, i32, i32, i32]),
    "w3d_knn_match_fwd": (i32, [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "w3d_knn_prepare": (i32, [vp, vp, i64, i64, i32, i32, i32, vp, vp, i32, vp, vp, sz, vp]),
    "w3d_knn_match_prepared": (i32, [vp, vp, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "w3d_knn_prepare_hybrid": (i32, [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, f32, vp, i32, vp, vp, sz, vp]),
    "w3d_knn_prune_plan": (i32, [vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, f32, vp, sz, vp]),
    "w3d_knn_prune_plan_lb": (i32, [vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, i32, vp, sz, vp]),
    "w3d_knn_prune_plan_err": (i32, [vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i32, vp, sz, vp]),
    "w3d_knn_match_prepared_lb": (i32, [vp, vp, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, sz, vp]),
    "w3d_knn_bounds_advance": (i32, [vp, i64, vp, vp]),
    "w3d_knn_bounds_reorder": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i64, vp]),
    "w3d_knn_plan_order": (i32, [i64, i64, i32, i32, vp, i32, vp, sz, vp]),
    "w3d_knn_prune_counts_offset": (sz, [i64, i64, i32, i32]),
    "w3d_knn_straggler_counts_offset": (sz, [i64, i64, i32, i32]),
    "w3d_knn_straggler_capacity": (i32, [i64, i64]),
    "w3d_knn_prepare_motion": (i32, [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "w3d_knn_chunk_norms": (i32, [vp, i64, i64, i32, i32, i32, i64, i64, i64, vp, sz, vp]),
    "w3d_knn_chunk_sweep": (i32, [vp, vp, i64, i64, i32, i32, i32, i32, i64, i64, i32, i32, vp, vp, vp, vp, sz, vp]),
    "w3d_knn_chunk_merge": (i32, [vp, vp, i64, i64, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "w3d_knn_loss": (i32, [vp, i64, i32, f32, vp, vp]),
    "w3d_knn_match_bwd": (i32, [vp, vp, vp, i64, i64, i32, i32, i32, f32, i32, vp, vp]),
    "w3d_knn_match_bwd_ordered": (i32, [vp, vp, vp, i64, i64, i32, i32, i32, f32, i32, vp, vp, vp]),
    "w3d_domain_loss_fwd_bwd": (i32, [vp, i64, i32, vp, vp, i32, i64, i32, f32, vp, vp, vp, vp]),
    "w3d_auxw_pairwise_fwd_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, vp, vp, vp]),
    "w3d_coverage_loss_bwd": (i32, [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp]),
    "w3d_sw_workspace_bytes": (sz, [i32, i32]),
    "w3d_sw_loss_fwd_bwd": (i32, [vp, vp, i32, i32, f32, vp, vp, vp, sz, vp]),
    "w3d_sw_max_points": (i32, []),
    "w3d_sw_project_sort": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "w3d_sw_features_workspace_bytes": (sz, [i32, i32]),
    "w3d_sw_features_fwd_bwd": (i32, [vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, sz, vp]),
    "w3d_emd2_matrix_fwd_bwd": (i32, [vp, i32, i32, i64, f32, vp, vp, vp, vp]),
    "w3d_emd2_points_fwd_bwd": (i32, [vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp]),
    "w3d_sqdist_matrix_fwd": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "w3d_sqdist_matrix_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "w3d_dual_entropic_fwd_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "w3d_dual_entropic_plan": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp]),
    "w3d_gram_workspace_bytes": (sz, [i32, i32]),
    "w3d_gram_matrix": (i32, [vp, i32, i32, vp, vp, sz, vp]),
    "w3d_gram_style_loss_fwd_bwd": (i32, [vp, vp, i32, i32, f32, vp, vp, vp, sz, vp]),
    "w3d_conv3x3": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "w3d_relu": (i32, [vp, i64, vp, vp]),
    "w3d_maxpool2": (i32, [vp, i32, i32, i32, vp, vp]),
    "w3d_relu_pool_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "w3d_nearest_resize": (i32, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "w3d_nearest_resize_bwd": (i32, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "w3d_knn_lists_workspace_bytes": (sz, [i64, i32]),
    "w3d_knn_lists": (i32, [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, sz, vp]),
    "w3d_radius_count": (i32, [vp, vp, i64, i64, i32, i64, f32, vp, vp]),
    "w3d_dist_row": (i32, [vp, vp, i64, i32, vp, vp]),
    "w3d_rowmax_dist": (i32, [vp, vp, i64, i64, i32, vp, vp]),
    "w3d_kmeans_assign": (i32, [vp, i64, i32, vp, i32, f64, vp, vp, vp, vp, vp, vp]),
    "w3d_kmeans_update": (i32, [vp, vp, i32, i32, f64, vp, vp, vp, vp]),
    "w3d_kmeanspp_pots": (i32, [vp, i64, i32, vp, i32, vp, vp, vp]),
    "w3d_kmeanspp_apply": (i32, [vp, i64, i32, vp, vp, vp]),
    "w3d_quantize_rows_fp8": (i32, [vp, i32, i64, i32, vp, vp, vp, vp]),
    "w3d_knn_fp8": (i32, [vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "w3d_knn_fp8_lb": (i32, [vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, sz, vp]),
    "w3d_knn_mfma_launches": (i32, [i64, i64]),
    "w3d_knn_mfma_repair_count": (i32, [vp, i64, i64, i32, vp, vp, vp]),
    "w3d_knn_mfma_repair_poll": (i32, [vp, i64, i64, i32, vp, vp, vp]),
    "w3d_knn_mfma_set_tuning": (i32, [i32, f32]),
    "w3d_knn_mfma_set_cu_reserve": (i32, [i32]),
    "w3d_knn_mfma_debug_clocks": (i32, [vp, vp]),
    "w3d_knn_mfma_debug_events": (i32, [vp, vp]),
    "w3d_list_max_fanin": (i32, [vp, i64, i32, i64, vp, vp, vp]),
    "w3d_absmax_16": (i32, [vp, i32, i64, vp, vp]),
    "w3d_descriptor_bwd_window_x4": (i32, [vp, vp, i64, i32, vp, i64, i32, i32, vp, vp]),
    "w3d_knn_descriptor_bwd_fused_x4": (i32, [vp, vp, i64, i32, i32, vp, vp, i64, f32, vp, i32, f32, i64, i32, i32, vp, vp, vp, vp, vp]),
    "w3d_knn_descriptor_bwd_fused_slots": (i32, [vp, vp, i64, i32, i32, vp, vp, i64, f32, vp, i32, f32, i64, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "w3d_knn_descriptor_bwd_fused_slots_cached": (i32, [vp, vp, i64, i32, i32, vp, vp, i64, f32, vp, i32, f32, i64, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "w3d_descriptor_fwd": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "w3d_pack_xyz4": (i32, [vp, i64, vp, vp]),
    "w3d_descriptor_fwd_x4": (i32, [vp, vp, i64, i32, i32, vp, vp, vp]),
    "w3d_descriptor_fwd_window_x4": (i32, [vp, i64, vp, i64, i32, i32, vp, vp, vp, i32, vp, vp]),
    "w3d_descriptor_fwd_slots_x4": (i32, [vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "w3d_descriptor_fwd_x4_motion": (i32, [vp, vp, i64, i32, i32, vp, vp, vp, vp, vp]),
    "w3d_descriptor_bwd_csr_x4": (i32, [vp, vp, i64, i32, vp, vp, vp, i64, vp, i32, vp, vp]),
    "w3d_descriptor_bwd": (i32, [vp, vp, i64, i32, vp, vp, vp]),
    "w3d_descriptor_vec_fwd": (i32, [vp, vp, vp, i64, i32, vp, vp]),
    "w3d_descriptor_vec_bwd": (i32, [vp, vp, vp, i64, i32, vp, vp, vp, vp]),
    "w3d_descriptor_mse_fwd_bwd": (i32, [vp, vp, vp, i64, i32, i64, f32, vp, vp, vp]),
    "w3d_csr_workspace_bytes": (sz, [i64, i64, i32]),
    "w3d_csr_build": (i32, [vp, i64, i32, i64, vp, vp, vp, sz, vp]),
    "w3d_descriptor_bwd_csr": (i32, [vp, vp, i64, i32, vp, vp, vp, i64, vp, i32, vp, vp]),
    "w3d_descriptor_mse_seg_csr": (i32, [vp, vp, vp, i64, i32, i64, i32, vp, vp, f32, vp, vp, vp, vp]),
    "w3d_adam_step": (i32, [C.POINTER(AdamGroup), i32, i32, C.c_double, C.c_double, C.c_double, i32, vp]),
    "w3d_tv_loss_fwd_bwd": (i32, [vp, i32, i32, i32, f32, i32, vp, vp, vp]),
    "w3d_mse_loss_fwd_bwd": (i32, [vp, vp, i64, f32, i32, vp, vp, vp]),
    "w3d_ssim_fwd": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "w3d_ssim_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, vp, vp]),
    "w3d_raster_geom_bytes": (sz, [i32]),
    "w3d_raster_image_bytes": (sz, [i32, i32]),
    "w3d_raster_binning_bytes": (sz, [i64]),
    "w3d_raster_forward": (i32, [C.POINTER(RasterIn), vp, vp, ALLOC_FN, vp, vp, vp, vp, C.POINTER(i32), vp]),
    "w3d_raster_backward": (i32, [C.POINTER(RasterIn), vp, C.POINTER(RasterGradIn), C.POINTER(RasterGradOut), vp, vp, vp, i32, vp]),
    "w3d_raster_backward_accumulate": (i32, [C.POINTER(RasterIn), vp, C.POINTER(RasterGradIn), C.POINTER(RasterGradOut), vp, vp, vp, i32, vp]),
    "w3d_raster_backward_selfclean": (i32, [C.POINTER(RasterIn), vp, C.POINTER(RasterGradIn), C.POINTER(RasterGradOut), vp, vp, vp, i32, i32, vp]),
    "w3d_raster_preprocess": (i32, [C.POINTER(RasterIn), vp, vp, vp]),
    "w3d_raster_band_scratch_bytes": (sz, [i32]),
    "w3d_raster_blend_forward": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, ALLOC_FN, vp, vp, vp, C.POINTER(i32), vp]),
    "w3d_raster_blend_backward": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "w3d_raster_preprocess_backward": (i32, [C.POINTER(RasterIn), vp, vp, C.POINTER(RasterGradOut), vp]),
    "w3d_mark_visible": (i32, [vp, vp, vp, i32, vp, vp]),
    "w3d_activations_fwd": (i32, [vp, vp, vp, i32, vp, vp, vp, vp]),
    "w3d_activations_bwd": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]),
}


class W3dError(RuntimeError):
    pass


def lib():
    """Load the native library (once).  Raises if it has not been built — there is no CPU fallback."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own HIP runtime (libamdhip64) under torch/lib; this library is linked against the system one.  Whichever is loaded
        # FIRST serves the whole process: if ours came first, torch and this library would sit on a runtime that torch did not initialise and every
        # launch here fails with "no ROCm-capable device is detected".  Import torch first, always.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise W3dError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(wast3d_amd has no fallback path without its HIP library)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError here = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, what: str = ""):
    if code != 0:
        detail = lib().w3d_last_hip_error_string().decode() if code == -2 else ""
        raise W3dError(f"{what or 'wast3d_hip call'} failed: {lib().w3d_strerror(code).decode()} ({code})" + (f": {detail}" if detail else ""))


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Zero-element tensors map to NULL like the reference's
    `.data<float>()` of an empty tensor (RAST/rasterize_points.cu)."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def clear_stale_hip_error():
    """The entry points report HIP's sticky per-thread error after their launches (W3D_CHECK_LAUNCH): an error some OTHER HIP user of this thread left behind
    (an event query's hipErrorNotReady, a failed probe) must not be charged to them with a misleading detail string.  Called at the head of every operator
    (require_cuda) and once per iteration by the loops — one argument-less native call."""
    lib().w3d_clear_hip_error()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise W3dError("wast3d_amd kernels need tensors on the MI355X (cuda device); got a CPU tensor. "
                           "There is deliberately no CPU path in the product package.")
    clear_stale_hip_error()
