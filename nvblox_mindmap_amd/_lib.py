"""ctypes binding of ``include/mmfusion.h`` (libmmfusion.so, HIP/gfx950).

This is the only place the package touches native code.  There is NO CPU fallback: if the
shared library is missing, or no HIP device is visible, the calls raise ``RuntimeError``.
Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C nvblox_mindmap_amd/csrc``.
"""
import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMF_LIB=<file name> selects another build in the package directory (e.g. libmmfusion_trace.so: the library with the
# per-workgroup timeline hooks, `make -C csrc WG_TRACE=1 OUT=../libmmfusion_trace.so BUILD=_build_trace`)
LIB_PATH = os.path.join(_HERE, os.environ.get("MMF_LIB", "libmmfusion.so"))
CSRC_DIR = os.path.join(_HERE, "csrc")

MMF_LAYER_TSDF, MMF_LAYER_COLOR, MMF_LAYER_FEATURE = 0, 1, 2
MMF_NUM_STATS = 9
KERNEL_IDS = {
    "raycast": 0, "alloc": 1, "tsdf": 2, "candidates": 3, "sphere": 4, "color": 5, "feature": 6, "decay": 7, "mesh": 8,
    "feature_flat": 9,
}


# the sources the FUSION kernels (the frame's launches, the mesh / model-input and image kernels) and their host side are built from
FUSION_SOURCES = ("mmf_kernels_map.hip", "mmf_kernels_app.hip", "mmf_kernels_mesh.hip", "mmf_kernels_image.hip", "mmf_api.hip",
                  "mmf_api_outputs.hip", "mmf_api_internal.h", "mmf_alloc_device.h", "mmf_app_device.h", "mmf_device.h", "mmf_mask_device.h",
                  "mmf_trace_device.h", "mmf_launch.h")


def source_hash(scope: str = "fusion") -> str:
    """16 hex digits identifying the native sources a build was made from.  ``scope="fusion"`` (the stamp of the counter summaries
    under profiles/: the frame's kernels are what they count): sha256 over FUSION_SOURCES + include/mmf_mc_table.h -- the policy
    side (FPS, the diffusion head's, the backbone's and the training step's kernels: mmf_kernels_{fps,policy*,backbone,train_*}.hip,
    mmf_api_ops.hip, mmf_launch_policy.h) and the ABI header's declarations do not change the fusion kernels' code.  ``scope="all"``:
    every csrc/*.hip, csrc/*.h and include/*.h.  bench.py compares a summary's stamp with the running tree's and marks replayed
    counters as stale when they differ (`roofline.counters_stale`)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    if scope == "all":
        files = sorted(glob.glob(os.path.join(CSRC_DIR, "*.hip")) + glob.glob(os.path.join(CSRC_DIR, "*.h"))
                       + glob.glob(os.path.join(os.path.dirname(_HERE), "include", "*.h")))
    else:
        files = sorted([os.path.join(CSRC_DIR, f) for f in FUSION_SOURCES] + [os.path.join(os.path.dirname(_HERE), "include", "mmf_mc_table.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class MmfParams(C.Structure):
    """``mmf_params`` of include/mmfusion.h (field order and types must match)."""

    _fields_ = [
        ("voxel_size_m", C.c_float),
        ("max_integration_distance_m", C.c_float),
        ("truncation_distance_vox", C.c_float),
        ("max_weight", C.c_float),
        ("weighting_mode", C.c_int32),
        ("lin_interp_max_diff_vox", C.c_float),
        ("appearance_measurement_weight", C.c_float),
        ("appearance_max_weight", C.c_float),
        ("raycast_subsampling", C.c_int32),
        ("workspace_bounds_type", C.c_int32),
        ("ws_min", C.c_float * 3),
        ("ws_max", C.c_float * 3),
        ("tsdf_decay_factor", C.c_float),
        ("decayed_weight_threshold", C.c_float),
        ("deallocate_decayed_blocks", C.c_int32),
        ("mesh_min_weight", C.c_float),
        ("st_subsampling", C.c_int32),
        ("st_max_steps", C.c_int32),
        ("st_max_ray_length_m", C.c_float),
        ("st_surface_eps_vox", C.c_float),
        ("feature_channels", C.c_int32),
        ("num_preallocated_blocks", C.c_int32),
        ("expansion_factor", C.c_float),
        ("raycast_to_truncation", C.c_int32),
        ("decay_appearance_layers", C.c_int32),
        ("raycast_walk_from_camera", C.c_int32),
        ("appearance_blend_division", C.c_int32),
        ("fma_contraction", C.c_int32),
        ("block_index_by_division", C.c_int32),
        ("view_truncation_band_marking", C.c_int32),
        ("bilinear_four_weight_sum", C.c_int32),
    ]


class MmfFrame(C.Structure):
    """``mmf_frame`` of include/mmfusion.h."""

    _fields_ = [
        ("struct_size", C.c_int),
        ("depth", C.c_void_p),
        ("rgb", C.c_void_p),
        ("features_f16", C.c_void_p),
        ("lowres_features", C.c_void_p),
        ("lowres_h", C.c_int), ("lowres_w", C.c_int), ("lowres_channels", C.c_int),
        ("input_mask", C.c_void_p),
        ("invert_input_mask", C.c_int),
        ("H", C.c_int), ("W", C.c_int), ("Hf", C.c_int), ("Wf", C.c_int), ("feature_channels", C.c_int),
        ("T_W_C", C.c_void_p),
        ("K", C.c_void_p),
        ("min_depth_m", C.c_float),
        ("input_mask_erosion_iterations", C.c_int), ("valid_depth_mask_erosion_iterations", C.c_int), ("border_percent", C.c_int),
        ("depth_mask_out", C.c_void_p),
        ("feature_mask_out", C.c_void_p),
    ]


# name -> (restype, argtypes): every symbol include/mmfusion.h declares
_VP, _I, _F = C.c_void_p, C.c_int, C.c_float
_PI = C.POINTER(C.c_int)
SIGNATURES = {
    "mmf_params_size": (_I, []),
    "mmf_abi_version": (_I, []),
    "mmf_default_params": (_I, [C.POINTER(MmfParams)]),
    "mmf_last_error": (C.c_char_p, []),
    "mmf_device_count": (_I, []),
    "mmf_mapper_create": (_I, [_I, C.POINTER(MmfParams), _I, C.POINTER(_VP)]),
    "mmf_mapper_destroy": (_I, [_VP]),
    "mmf_num_mappers": (_I, [_VP]),
    "mmf_add_depth_frame": (_I, [_VP, _I, _VP, _VP, _I, _I, _VP, _VP, _VP]),
    "mmf_add_color_frame": (_I, [_VP, _I, _VP, _VP, _I, _I, _VP, _VP, _VP]),
    "mmf_add_feature_frame": (_I, [_VP, _I, _VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "mmf_integrate_frame": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _VP, _F, _I, _I, _I, _VP, _VP, _VP]),
    "mmf_add_feature_frame_lowres": (_I, [_VP, _I, _VP, _I, _I, _I, _VP, _I, _I, _VP, _VP, _VP]),
    "mmf_integrate_frame_lowres": (_I, [_VP, _I, _VP, _VP, _VP, _I, _I, _I, _VP, _I, _I, _I, _I, _VP, _VP, _F, _I, _I, _I, _VP, _VP,
                                         _VP]),
    "mmf_integrate_frame_batch": (_I, [_I, C.POINTER(_VP), C.POINTER(_I), C.POINTER(MmfFrame), _VP]),
    "mmf_integrate_frame_desc": (_I, [_VP, _I, C.POINTER(MmfFrame), _VP]),
    "mmf_decay": (_I, [_VP, _I, _VP]),
    "mmf_set_deferred_feature_rows": (_I, [_VP, _I, _I]),
    "mmf_flush": (_I, [_VP, _I, _VP]),
    "mmf_deferred_feature_rows_pending": (_I, [_VP, _I]),
    "mmf_clear": (_I, [_VP, _I, _VP]),
    "mmf_update_feature_mesh": (_I, [_VP, _I, _VP, _PI]),
    "mmf_get_feature_mesh": (_I, [_VP, _I, _VP, _VP, _VP]),
    "mmf_model_inputs_prepare": (_I, [_VP, _I, _VP, _VP, _I, _I, _VP, _PI]),
    "mmf_model_inputs_gather": (_I, [_VP, _I, _VP, _I, _I, _VP, _VP, _I, _VP, _VP]),
    "mmf_host_randperm_prefix": (_I, [_VP, C.c_int64, C.c_int64, C.c_int64, _VP]),
    "mmf_host_read_file_at": (_I, [C.c_char_p, C.c_int64, _VP, C.c_int64]),
    "mmf_host_sample_vertex_file": (_I, [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _VP, C.c_int64, _VP, _VP]),
    "mmf_num_allocated_blocks": (_I, [_VP, _I, _I, _VP, _PI]),
    "mmf_get_block_indices": (_I, [_VP, _I, _I, _VP, _I, _VP]),
    "mmf_get_tsdf_blocks": (_I, [_VP, _I, _VP, _I, _VP]),
    "mmf_get_feature_blocks": (_I, [_VP, _I, _VP, _VP, _I, _VP]),
    "mmf_get_color_blocks": (_I, [_VP, _I, _VP, _VP, _I, _VP]),
    "mmf_update_mesh_topology": (_I, [_VP, _I, _VP, _PI, _PI]),
    "mmf_get_mesh_topology": (_I, [_VP, _I, _VP, _VP, _VP]),
    "mmf_import_blocks": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I, _VP]),
    "mmf_query_layer": (_I, [_VP, _I, _I, _VP, _I, _VP, _VP]),
    "mmf_backproject_depth": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP]),
    "mmf_sample_inputs_scratch_floats": (_I, []),
    "mmf_sample_frame_inputs": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mmf_sample_frame_inputs_host": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mmf_erode_mask": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmf_feature_mask": (_I, [_VP, _VP, _I, _I, _F, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "mmf_depth_mask": (_I, [_VP, _VP, _I, _I, _F, _VP, _VP]),
    "mmf_frame_masks": (_I, [_VP, _VP, _I, _I, _F, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "mmf_upsample_features": (_I, [_VP, _I, _I, _I, _VP, _I, _I, _I, _VP]),
    "mmf_upsample_features_spec": (_I, [_VP, _I, _I, _I, _VP, _I, _I, _I, _I, _VP]),
    "mmf_rotary_apply": (_I, [_VP, C.c_longlong, _VP, _VP, _VP, C.c_longlong, _I, _VP]),
    "mmf_rotary_apply_grad": (_I, [_VP, _VP, _VP, _VP, C.c_longlong, _I, _VP]),
    "mmf_adaln_modulate": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmf_qkv_block": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmf_out_ffn_block": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _I, _I, _I, _VP]),
    "mmf_qkv_heads": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "mmf_attention_heads": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "mmf_out_ffn_mfma": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _I, _I, _I, _VP]),
    "mmf_step_prologue": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, C.c_longlong, _I, _VP]),
    "mmf_head_outputs": (_I, [_VP, _VP, C.c_longlong, _I, _I, _I, _VP, _VP, _VP, _I, _VP]),
    "mmf_step_tail": (_I, [_VP, _VP, C.c_longlong, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                           C.c_longlong, _I, _VP]),
    "mmf_out_ffn_qkv": (_I, [_VP, _F, _F, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _I, _VP]),
    "mmf_debug_wg_trace": (_I, [_VP, _I]),
    "mmf_qkv_heads2": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "mmf_out_ffn_mfma2": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmf_out_ffn_qkv2": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "mmf_split_linear_weight": (_I, [_VP, _I, _I, _VP, _VP]),
    "mmf_split_activations3": (_I, [_VP, C.c_int64, _I, _VP, _VP]),
    "mmf_attention_split": (_I, [_VP, _VP, _VP, C.c_int64, C.c_int64, _I, _I, _I, _I, _F, _VP, _I, _VP]),
    "mmf_adaln_modulate_grad_scratch_bytes": (C.c_int64, [_I]),
    "mmf_adaln_modulate_grad": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "mmf_linear_weight_grad_scratch_bytes": (C.c_int64, [C.c_int64, _I, _I]),
    "mmf_linear_weight_grad": (_I, [_VP, _VP, C.c_int64, _I, _I, _VP, _VP, _VP, _VP]),
    "mmf_layernorm_train_scratch_bytes": (C.c_int64, []),
    "mmf_layernorm_train_forward": (_I, [_VP, _VP, _VP, _VP, _F, C.c_int64, _I, _VP, _VP, _VP, _VP, _VP]),
    "mmf_layernorm_train_backward": (_I, [_VP, _VP, _VP, _VP, _VP, C.c_int64, _I, _VP, _VP, _VP, _VP, _VP]),
    "mmf_train_attention_forward": (_I, [_VP, _VP, _VP, C.POINTER(C.c_int64), _VP, _I, _I, _I, _I, _I, _F, _VP, _VP, _VP]),
    "mmf_train_attention_backward": (_I, [_VP, _VP, _VP, C.POINTER(C.c_int64), _VP, _I, _I, _I, _I, _I, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mmf_gelu_split_activations3": (_I, [_VP, C.c_int64, _I, _VP, _VP]),
    "mmf_split_attention_heads3": (_I, [_VP, C.c_int64, _I, _I, _I, _VP, _VP]),
    "mmf_layernorm_split_activations3": (_I, [_VP, _VP, _VP, _VP, _F, C.c_int64, _I, _VP, _VP, _VP]),
    "mmf_self_layer": (_I, [_VP, C.c_float, C.c_float, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_uint32, _I, _I, _I, _I, _VP]),
    "mmf_cross_layer": (_I, [_VP, C.c_float, C.c_float, _VP, _VP, _VP, _VP, _VP, _VP, C.c_uint32, _I, _I, _I, _I, _I, _VP]),
    "mmf_attention_heads_split": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, C.POINTER(C.c_int), _VP]),
    "mmf_out_ffn_mfma_partials": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _I, _I, _I, _VP]),
    "mmf_ffn_block": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _I, _I, _I, _VP]),
    "mmf_q_block": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmf_kv_block": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_longlong, _I, _VP]),
    "mmf_attn_out_block": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, C.c_longlong, _I, _VP]),
    "mmf_ddpm_step": (_I, [_VP, _VP, C.c_longlong, _VP, _VP, C.c_longlong, _I, _I, _VP, _VP, _VP]),
    "mmf_attention_small": (_I, [_VP, _VP, C.c_longlong, _VP, C.c_longlong, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "mmf_farthest_point_sampling": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP]),
    "mmf_fps_workspace_bytes": (C.c_int64, [_I, _I, _I]),
    "mmf_farthest_point_sampling_ws": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP, C.c_int64, _VP]),
    "mmf_get_synthetic_depth_dims": (_I, [_VP, _I, _PI, _PI]),
    "mmf_get_synthetic_depth": (_I, [_VP, _I, _VP, _VP]),
    "mmf_render_synthetic_depth": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP]),
    "mmf_last_view_block_count": (_I, [_VP, _I, _VP, _PI]),
    "mmf_get_last_view_blocks": (_I, [_VP, _I, _VP, _I, _VP]),
    "mmf_get_stats": (_I, [_VP, _I, _VP, C.POINTER(C.c_int64)]),
    "mmf_reset_stats": (_I, [_VP, _I, _VP]),
    "mmf_integrate_frame_multi": (_I, [_VP, _I, _PI, C.POINTER(MmfFrame), _VP]),
    "mmf_get_alloc_timeline": (_I, [_VP, _I, _I, C.POINTER(C.c_int64)]),
    "mmf_debug_alloc_recoveries": (_I, [_VP, _I, _VP, C.POINTER(C.c_int64)]),
    "mmf_debug_hash_state": (_I, [_VP, _I, _I, _VP, C.POINTER(C.c_int64)]),
    "mmf_debug_count_tombstones": (_I, [_VP, _I, _I, _VP, C.POINTER(C.c_int64)]),
    "mmf_profile_enable": (_I, [_VP, _I]),
    "mmf_profile_set_stride": (_I, [_VP, _I]),
    "mmf_profile_get": (_I, [_VP, _I, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mmf_profile_reset": (_I, [_VP]),
    "mmf_kernel_name": (C.c_char_p, [_I]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile libmmfusion.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC_DIR, "-j4", "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce " + LIB_PATH)
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `make -C nvblox_mindmap_amd/csrc` or `__graft_entry__.build()`); there is no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.mmf_params_size() != C.sizeof(MmfParams):
            raise RuntimeError("mmf_params layout mismatch between the binding and libmmfusion.so")
        _lib = L
    return _lib


def last_error() -> str:
    return lib().mmf_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    """Raise RuntimeError for a non-zero return code (the reference raises from C++ asserts)."""
    if rc != 0:
        raise RuntimeError(f"libmmfusion {what} failed (code {rc}): {last_error()}")


def default_params() -> MmfParams:
    p = MmfParams()
    check(lib().mmf_default_params(C.byref(p)), "mmf_default_params")
    return p


def device_count() -> int:
    return lib().mmf_device_count()


def require_gpu() -> None:
    if device_count() <= 0:
        raise RuntimeError("no HIP device visible: nvblox_mindmap_amd runs on MI355X only (no CPU fallback)")


def stream_ptr(device: Optional[int] = None) -> C.c_void_p:
    """hipStream_t of torch's current stream on `device` (None: the current device; an index; a torch.device)."""
    import torch

    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:  # the per-frame paths ask several times per call: the raw handle without the Stream object (0.3 us vs 4 us)
        if device is None:
            idx = torch.cuda.current_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
        return C.c_void_p(raw(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dptr(t) -> C.c_void_p:
    """Device pointer of a torch tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())
