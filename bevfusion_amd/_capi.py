"""ctypes binding of libbevfusion_amd.so (C ABI declared in include/bevfusion_amd.h).

This is the only place the shared library is loaded.  There is NO fallback: if the
library is missing or a symbol cannot be resolved, importing a product op raises.
PyTorch is used by callers for device memory and streams only.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BEVAMD_LIB") or os.path.join(_HERE, "lib", "libbevfusion_amd.so")   # BEVAMD_LIB: tools/exp_build.sh A/B builds
EXT_LIB_PATH = os.path.join(_HERE, "lib", "libbevfusion_amd_ext.so")

_lib = None

P = c_void_p
I = c_int
Z = c_size_t
LL = c_longlong

# name -> (restype, argtypes); must list every function of include/bevfusion_amd.h
_SIGNATURES = {
    "bevamd_last_error": (c_char_p, []),
    # bev_pool
    "bevamd_bev_pool_forward": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_forward_bf16": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_backward": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_prepare_workspace_bytes": (Z, [I, I, I, I, I]),
    "bevamd_bev_pool_prepare": (I, [P, I, I, I, I, I, I, P, P, P, P, P, P, P, P, Z, P]),
    "bevamd_bev_pool_prepare_from_geom": (I, [P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, Z, P]),
    "bevamd_bev_pool_forward_cells": (I, [P, I, P, P, P, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_forward_cells_tuned": (I, [P, I, P, P, P, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_fused_forward": (I, [P, P, I, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_fused_schedule_workspace_bytes": (Z, [I]),
    "bevamd_bev_pool_fused_schedule": (I, [P, P, I, I, I, I, I, I, I, I, P, P, P, Z, P]),
    "bevamd_bev_pool_fused_forward_scheduled": (I, [P, P, I, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_fused_columns_supported": (I, [I, I, I, I]),
    "bevamd_bev_pool_fused_backward_columns_supported": (I, [I, I, I, I]),
    "bevamd_bev_pool_fused_columns_occupancy": (I, [I, I, I, I]),
    "bevamd_bev_pool_fused_columns_lds_bytes": (Z, [I, I, I, I]),
    "bevamd_bev_pool_fused_columns_workspace_bytes": (Z, [I, I]),
    "bevamd_bev_pool_fused_columns_count": (I, [P, I, I, I, I, I, I, I, I, P, P, P, P, P, Z, P]),
    "bevamd_bev_pool_fused_columns_build": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P, P, P, Z, P]),
    "bevamd_bev_pool_fused_forward_columns": (I, [P, P, I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_fused_backward_columns": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_cell_of_point": (I, [P, P, I, P, P]),
    "bevamd_bev_pool_fused_backward": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_backward_rows": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "bevamd_bev_pool_backward_points": (I, [P, P, P, I, I, I, I, I, I, P]),
    # training-mode BatchNorm over sparse rows
    "bevamd_sparse_bn_workspace_bytes": (Z, [I]),
    "bevamd_sparse_bn_stats": (I, [P, I, LL, I, LL, c_float, c_float, P, P, P, P, P, Z, P]),
    "bevamd_sparse_bn_apply": (I, [P, I, LL, I, LL, P, P, P, P, P, LL, I, P, LL, P]),
    "bevamd_sparse_bn_backward": (I, [P, LL, P, LL, P, LL, I, LL, I, I, P, P, P, P, P, P, LL, P, LL, P, Z, P]),
    # view-transform glue
    "bevamd_depth_raster_workspace_bytes": (Z, [I, I, I]),
    "bevamd_depth_raster": (I, [P, I, I, P, P, P, P, I, I, I, P, P, Z, P]),
    "bevamd_depth_raster_batch": (I, [P, P, I, I, P, P, I, P, P, I, I, I, P, P, Z, P]),
    "bevamd_depth_raster_batch_zero_ws": (I, [P, P, I, I, P, P, I, P, P, I, I, I, P, P, Z, P]),
    "bevamd_lss_geometry": (I, [P, I, P, P, P, P, P, P, I, I, P, P]),
    "bevamd_mat3_inverse": (I, [P, LL, LL, I, P, P]),
    "bevamd_mat3_inverse_with_column": (I, [P, LL, LL, I, P, P, P]),
    "bevamd_lss_camera_matrices": (I, [P, P, P, LL, LL, I, P, P, P]),
    # voxelization
    "bevamd_hard_voxelize_workspace_bytes": (Z, [I]),
    "bevamd_hard_voxelize": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, Z, P]),
    "bevamd_dynamic_voxelize": (I, [P, P, P, P, I, I, I, P]),
    "bevamd_voxelize_mean": (I, [P, P, P, P, P, P, I, I, I, I, I, P, P, Z, P]),
    "bevamd_voxel_compact": (I, [P, P, P, P, I, I, I, P, P, P, P, P]),
    "bevamd_voxelize_mean_batch_workspace_bytes": (Z, [P, I]),
    "bevamd_voxelize_mean_batch": (I, [P, P, I, I, P, P, I, I, I, P, P, P, P, P, P, Z, P]),
    "bevamd_voxelize_mean_batch_ex": (I, [P, P, I, I, P, P, I, I, I, I, P, P, P, P, P, P, Z, P]),
    "bevamd_voxelize_mean_batch_rows16": (I, [P, P, I, I, P, P, I, I, I, I, P, P, P, P, P, P, I, I, P, Z, P]),
    # spconv
    "bevamd_spconv_rulebook_workspace_bytes": (Z, [I, I, P, I]),
    "bevamd_spconv_hash_index_bytes": (Z, [I]),
    "bevamd_spconv_rank_index_bytes": (Z, [I, P]),
    "bevamd_spconv_hash_index_build": (I, [P, I, P, I, P, P, Z, P]),
    "bevamd_spconv_downsample": (I, [P, I, P, I, P, P, P, P, P, P, I, P, P, Z, P, I, P]),
    "bevamd_spconv_downsample_sorted": (I, [P, I, P, I, P, P, P, P, P, I, P, P, I, P, P, Z, P]),
    "bevamd_spconv_neighbors": (I, [P, I, P, I, P, P, P, P, P, I, I, P, I, P, I, P]),
    "bevamd_spconv_max_outputs": (I, [I, P, P, I]),
    "bevamd_spconv_build_rulebook": (I, [P, I, I, P, P, P, P, P, P, I, I, P, I, P, I, P, P, P, Z, P]),
    "bevamd_spconv_max_outputs_ex": (I, [I, P, P, P, I, I]),
    "bevamd_spconv_dense_bev": (I, [P, I, I, I, I, P, I, I, P, P, P]),
    "bevamd_spconv_pairs_workspace_bytes": (Z, [I, I]),
    "bevamd_spconv_pairs_from_nbr": (I, [P, I, I, I, P, I, P, P, Z, P]),
    "bevamd_spconv_nbr_from_pairs": (I, [P, I, P, I, I, P, I, P]),
    "bevamd_spconv_transpose_nbr": (I, [P, I, I, I, P, I, P]),
    "bevamd_spconv_prepared_filter_elems": (Z, [I, I, I, I, I]),
    "bevamd_spconv_prepare_filters": (I, [P, I, I, I, I, I, P, P]),
    "bevamd_spconv_conv_forward": (I, [P, I, P, P, I, I, P, I, I, I, P, P, P, P, P, I, P]),
    "bevamd_spconv_tiled_supported": (I, [I, I, I]),
    "bevamd_spconv_filter_image_elems": (Z, [I, I, I, I]),
    "bevamd_spconv_make_filter_image": (I, [P, I, I, I, I, I, P, P]),
    "bevamd_spconv_make_filter_images": (I, [I, P, P, P, P, P, P, I, P]),
    "bevamd_spconv_conv_forward_tiled": (I, [P, I, I, I, P, P, I, I, P, I, I, I, P, I, P, P, P, P, I, I, I, P]),
    "bevamd_spconv_f32x3_supported": (I, [I, I]),
    "bevamd_spconv_filter_image3_elems": (Z, [I, I, I, I]),
    "bevamd_spconv_make_filter_image3": (I, [P, I, I, I, I, P, P]),
    "bevamd_spconv_conv_forward_f32x3": (I, [P, I, I, P, P, I, I, P, I, I, I, P, I, P, P, P, P, I, I, P]),
    "bevamd_spconv_conv_forward_tiled_slots": (I, [P, I, I, I, P, P, P, I, I, P, I, I, P, I, P, P, P, P, I, I, I, P]),
    "bevamd_spconv_pad_cast_rows": (I, [P, I, I, I, I, P, P]),
    "bevamd_spconv_slab_set_profile_buffer": (None, [P]),
    "bevamd_spconv_slab_ablation_mask": (I, []),
    "bevamd_spconv_slab_block_rows": (I, [I, I]),
    "bevamd_spconv_slab_variants": (I, [I, P, I]),
    "bevamd_spconv_slab_grid_ok": (I, [P, I]),
    "bevamd_spconv_slab_hdr_bytes": (Z, [I, I]),
    "bevamd_spconv_slab_slot_bytes": (Z, [I, I]),
    "bevamd_spconv_slab_build": (I, [P, I, I, P, I, P, P, P, P]),
    "bevamd_spconv_slab_build_from_index": (I, [P, I, P, I, P, I, P, I, I, P, P, P, P]),
    "bevamd_spconv_sorted_index_bytes": (Z, [I, I, P]),
    "bevamd_spconv_sorted_index_build": (I, [P, I, P, I, P, P, Z, P, P]),
    "bevamd_spconv_slab_build_from_sorted": (I, [P, I, P, I, P, P, P, P, I, P, I, I, P, P, P, P]),
    "bevamd_spconv_conv_forward_slab": (I, [P, I, I, I, P, P, P, I, I, P, I, I, P, I, P, P, P, P, I, I, I, P]),
    "bevamd_spconv_wgrad_workspace_bytes": (Z, [I, I, I]),
    "bevamd_spconv_conv_wgrad": (I, [P, P, I, P, I, I, I, I, I, P, P, Z, P]),
    "bevamd_spconv_wgrad_slab_set_profile_buffer": (None, [P]),
    "bevamd_spconv_wgrad_slab_supported": (I, [I, I, I]),
    "bevamd_spconv_wgrad_slab_block_rows": (I, [I]),
    "bevamd_spconv_wgrad_slab_workspace_bytes": (Z, [I, I]),
    "bevamd_spconv_conv_wgrad_slab": (I, [P, I, I, P, I, I, P, P, I, I, I, I, P, P, Z, P]),
    # iou3d
    "bevamd_iou3d_boxes_overlap_bev": (I, [P, I, P, I, P, P]),
    "bevamd_iou3d_boxes_iou_bev": (I, [P, I, P, I, P, P]),
    "bevamd_iou3d_nms_workspace_bytes": (Z, [I]),
    "bevamd_iou3d_nms": (I, [P, I, c_float, I, P, P, P, P, Z, P]),
    # primitives
    "bevamd_scan_workspace_bytes": (Z, [Z]),
    "bevamd_exclusive_scan_u32": (I, [P, P, Z, P, P, Z, P]),
    "bevamd_scan_single_pass_state_bytes": (Z, [Z]),
    "bevamd_exclusive_scan_u32_single_pass": (I, [P, P, Z, P, P, Z, P]),
    "bevamd_radix_sort_workspace_bytes": (Z, [Z]),
    "bevamd_radix_sort_pairs_u32": (I, [P, P, P, P, Z, I, P, Z, P]),
    "bevamd_radix_sort_segmented_workspace_bytes": (Z, [P, I]),
    "bevamd_radix_sort_segmented_lanes": (I, [P, I, P, P]),
    "bevamd_debug_lds_poison": (I, [ctypes.c_uint32, P]),
    "bevamd_radix_sort_pairs_u32_segmented": (I, [P, P, P, P, P, I, I, P, Z, P]),
}


# libbevfusion_amd_ext.so (include/bevfusion_amd_ext.h): exports of the reference's pybind modules outside the hot path
# exported by -DBEVAMD_PROFILING builds only (rejected experiments kept for sweeps): bound when present
_PROFILING_SIGNATURES = {
    "bevamd_bev_pool_striped_line": (I, [I, I, I, I, P, I]),
}

_EXT_SIGNATURES = {
    "bevamd_dynamic_scatter_workspace_bytes": (Z, [I]),
    "bevamd_dynamic_scatter_index": (I, [P, I, I, P, P, P, P, P, P, P, P, Z, P]),
    "bevamd_dynamic_scatter_reduce": (I, [P, I, P, P, I, I, P, P]),
    "bevamd_dynamic_scatter_backward": (I, [P, P, P, P, P, P, I, I, I, I, P, P]),
    "bevamd_spconv_maxpool_forward": (I, [P, I, I, P, I, I, I, I, P, I, P]),
    "bevamd_spconv_maxpool_backward": (I, [P, P, P, I, P, I, I, I, I, P, P]),
}


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library handle; raise loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU/PyTorch fallback). "
            "Build it with `python -m bevfusion_amd.build` or `__graft_entry__.build()`."
        )
    # PyTorch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch FIRST so that the
    # dynamic loader binds this library to the runtime that owns torch's device memory and streams;
    # loading ours first would put two HIP runtimes in the process ("no ROCm-capable device").
    import torch  # noqa: F401

    lib = _bind(LIB_PATH, _SIGNATURES, ctypes.RTLD_GLOBAL)   # global: the optional ext library resolves its shared primitives here
    _lib = _Library(lib)
    return _lib


def _bind(path, signatures, mode=ctypes.DEFAULT_MODE):
    lib = ctypes.CDLL(path, mode=mode)
    for name, (res, args) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise NativeLibraryMissing(f"symbol {name} missing from {path}: rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    if signatures is _SIGNATURES:
        for name, (res, args) in _PROFILING_SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
    return lib


class _Library:
    """The hot-path library, plus — on first use of one of its symbols — the optional ext library (sparse max pooling, dynamic
    scatter: OUT of SURVEY.md §8's path, built into libbevfusion_amd_ext.so).  Callers see one namespace."""

    def __init__(self, main):
        self._main, self._ext = main, None

    def __getattr__(self, name):
        if name in _EXT_SIGNATURES:
            if self._ext is None:
                if not os.path.exists(EXT_LIB_PATH):
                    raise NativeLibraryMissing(f"{EXT_LIB_PATH} not found: {name} lives in the optional ext library "
                                               "(`python -m bevfusion_amd.build` builds both)")
                self._ext = _bind(EXT_LIB_PATH, _EXT_SIGNATURES)
            return getattr(self._ext, name)
        return getattr(self._main, name)


def exported_names():
    return list(_SIGNATURES)


def ext_exported_names():
    return list(_EXT_SIGNATURES)


def last_error():
    return load().bevamd_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"bevfusion_amd {what} failed (code {rc}): {last_error()}")


def ptr(t):
    """Device (or host) address of a tensor, None -> NULL."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """The HIP stream PyTorch is currently using on `device`, as void*."""
    import torch

    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def float3(values):
    arr = (c_float * 3)(*[float(v) for v in values])
    return arr


def ints(values):
    vals = [int(v) for v in values]
    return (c_int * len(vals))(*vals)


def floats(values):
    vals = [float(v) for v in values]
    return (c_float * len(vals))(*vals)
