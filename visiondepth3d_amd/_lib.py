"""ctypes binding of libvd3d_hip.so (include/vd3d.h).  There is NO fallback path: if the HIP library
is missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

from . import _abi
from ._abi import FrameScalars, RenderParams, ShiftParams, State

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VD3D_LIB_PATH") or os.path.join(_HERE, "libvd3d_hip.so")   # override: A/B builds of the same ABI

# every symbol include/vd3d.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = (
    "vd3d_abi_version", "vd3d_last_error", "vd3d_shift_params_default", "vd3d_render_params_default",
    "vd3d_ctx_create", "vd3d_ctx_destroy", "vd3d_sync", "vd3d_ctx_stream", "vd3d_ctx_set_stream", "vd3d_ctx_pixel_stream", "vd3d_ctx_pixel_stream_k",
    "vd3d_state_reset", "vd3d_state_new_clip", "vd3d_state_export", "vd3d_state_import", "vd3d_state_planes",
    "vd3d_last_scalars", "vd3d_pixel_shift", "vd3d_render_frame", "vd3d_render_frame_blank", "vd3d_depth_handoff", "vd3d_heal_missing_pixels", "vd3d_preview_heatmap", "vd3d_preview_arrows", "vd3d_conv3x3_c64_f16", "vd3d_conv3x3_head_f16", "vd3d_esr_tail_f32", "vd3d_nv12_to_bgr", "vd3d_bgr_to_nv12", "vd3d_resize_cubic_u8", "vd3d_resize_area_u8", "vd3d_resize_linear_u8", "vd3d_format_3d_output", "vd3d_esr_preprocess", "vd3d_esr_postprocess", "vd3d_add_weighted_u8", "vd3d_rife_preprocess", "vd3d_rife_postprocess",
    "vd3d_shard_begin", "vd3d_shard_pixels", "vd3d_shard_pixels_blank", "vd3d_tdf_plane_export", "vd3d_tdf_plane_import", "vd3d_set_pixel_overlap", "vd3d_join_pixels", "vd3d_wait_pixels", "vd3d_finish_frame", "vd3d_quantiles",
    "vd3d_subject_depth", "vd3d_shard2_p0", "vd3d_shard2_set_crops", "vd3d_shard2_p1", "vd3d_shard2_p1_batch", "vd3d_shard2_r1", "vd3d_shard2_p3", "vd3d_shard2_p3_batch", "vd3d_shard2_r2", "vd3d_depth_preprocess", "vd3d_add_layernorm", "vd3d_gemm_x3_weight_bytes", "vd3d_gemm_x3_pack_weights", "vd3d_gemm_x3", "vd3d_attention_x3_workspace_bytes", "vd3d_attention_x3", "vd3d_conv3x3_x2_weight_bytes", "vd3d_conv3x3_x2_pack_weights", "vd3d_conv3x3_x2", "vd3d_upsample_bilinear_nhwc", "vd3d_nhwc_bias_act_f32", "vd3d_upsample_bilinear_bias_nhwc_f32", "vd3d_dpt_head_tail_f32", "vd3d_preview_image", "vd3d_detect_black_bars", "vd3d_stream_copy", "vd3d_torch_math", "vd3d_torch_math_aten", "vd3d_debug_gaussian_kernel1d", "vd3d_debug_exp_torch", "vd3d_set_profiling", "vd3d_last_stage_ms", "vd3d_stage_calls",
    "vd3d_debug_planes", "vd3d_debug_tune",
)

_lib = None


class Vd3dError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvd3d_hip error {code}: {msg}")
        self.code = code


def lib():
    """Load libvd3d_hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C visiondepth3d_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, f32p, u8p = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p
    L.vd3d_abi_version.restype = i32
    L.vd3d_last_error.restype = C.c_char_p
    L.vd3d_ctx_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.vd3d_ctx_destroy.argtypes = [vp]
    L.vd3d_sync.argtypes = [vp]
    L.vd3d_ctx_stream.argtypes = [vp]
    L.vd3d_ctx_stream.restype = vp
    L.vd3d_ctx_set_stream.argtypes = [vp, vp]
    L.vd3d_ctx_pixel_stream.argtypes = [vp]
    L.vd3d_ctx_pixel_stream.restype = vp
    L.vd3d_ctx_pixel_stream_k.argtypes = [vp, i32]
    L.vd3d_ctx_pixel_stream_k.restype = vp
    for n in ("vd3d_state_reset", "vd3d_state_new_clip"):
        getattr(L, n).argtypes = [vp]
    L.vd3d_state_export.argtypes = [vp, C.POINTER(State)]
    L.vd3d_state_import.argtypes = [vp, C.POINTER(State)]
    L.vd3d_state_planes.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    L.vd3d_last_scalars.argtypes = [vp, C.POINTER(FrameScalars)]
    L.vd3d_pixel_shift.argtypes = [vp, f32p, f32p, i32, i32, i32, i32, C.POINTER(ShiftParams), u8p, u8p, f32p]
    L.vd3d_render_frame.argtypes = [vp, u8p, vp, i32, C.POINTER(RenderParams), u8p]
    L.vd3d_render_frame_blank.argtypes = [vp, u8p, vp, i32, C.POINTER(RenderParams), u8p]
    L.vd3d_shard_begin.argtypes = [vp, C.POINTER(RenderParams), i32]
    L.vd3d_shard_pixels.argtypes = [vp, i32, C.POINTER(RenderParams), u8p]
    L.vd3d_shard_pixels_blank.argtypes = [vp, i32, u8p, C.POINTER(RenderParams), u8p]
    L.vd3d_tdf_plane_export.argtypes = [vp, f32p, i32, i32]
    L.vd3d_tdf_plane_import.argtypes = [vp, f32p, i32, i32, i32]
    L.vd3d_set_pixel_overlap.argtypes = [vp, i32]
    L.vd3d_join_pixels.argtypes = [vp]
    L.vd3d_wait_pixels.argtypes = [vp, i32]
    L.vd3d_depth_handoff.argtypes = [vp, f32p, i32, i32, i32, i32, i32, i32, u8p]
    L.vd3d_heal_missing_pixels.argtypes = [vp, f32p, f32p, f32p, i32, i32, C.c_double, f32p]
    L.vd3d_resize_cubic_u8.argtypes = [vp, u8p, i32, i32, i32, u8p, i32, i32]
    L.vd3d_nv12_to_bgr.argtypes = [vp, u8p, C.c_longlong, u8p, C.c_longlong, i32, i32, u8p]
    L.vd3d_bgr_to_nv12.argtypes = [vp, u8p, i32, i32, u8p, C.c_longlong, u8p, C.c_longlong]
    L.vd3d_conv3x3_c64_f16.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    L.vd3d_conv3x3_head_f16.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    L.vd3d_esr_tail_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.vd3d_preview_arrows.argtypes = [vp, u8p, f32p, i32, i32, u8p]
    L.vd3d_preview_heatmap.argtypes = [vp, i32, f32p, i32, i32, u8p, u8p]
    L.vd3d_resize_area_u8.argtypes = [vp, u8p, i32, i32, u8p, i32, i32]
    L.vd3d_resize_linear_u8.argtypes = [vp, u8p, i32, i32, u8p, i32, i32]
    L.vd3d_format_3d_output.argtypes = [vp, u8p, u8p, i32, i32, i32, u8p]
    L.vd3d_esr_preprocess.argtypes = [vp, i32, u8p, C.c_longlong, i32, i32, i32, vp]
    L.vd3d_esr_postprocess.argtypes = [vp, f32p, i32, i32, i32, i32, i32, i32, i32, u8p, C.c_longlong]
    L.vd3d_rife_preprocess.argtypes = [vp, i32, u8p, u8p, i32, i32, i32, vp]
    L.vd3d_rife_postprocess.argtypes = [vp, f32p, i32, i32, i32, u8p]
    L.vd3d_add_weighted_u8.argtypes = [vp, u8p, C.c_double, u8p, C.c_double, C.c_double, C.c_longlong, u8p]
    L.vd3d_finish_frame.argtypes = [vp, u8p, u8p, f32p, i32, i32, C.POINTER(RenderParams), C.c_double, i32, i32, u8p]
    L.vd3d_quantiles.argtypes = [vp, f32p, C.c_int64, C.POINTER(C.c_float), i32, C.POINTER(C.c_float)]
    L.vd3d_subject_depth.argtypes = [vp, f32p, i32, i32, C.POINTER(C.c_float)]
    L.vd3d_shard2_p0.argtypes = [vp, vp, vp, vp]
    L.vd3d_shard2_set_crops.argtypes = [vp, vp, i32]
    L.vd3d_shard2_p1.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp]
    L.vd3d_shard2_p1_batch.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, vp]
    L.vd3d_shard2_p3_batch.argtypes = [vp, i32, i32, i32, vp, vp]
    L.vd3d_shard2_r1.argtypes = [vp, vp, i32]
    L.vd3d_shard2_p3.argtypes = [vp, i32, i32, vp, vp]
    L.vd3d_shard2_r2.argtypes = [vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_uint8), i32, vp]
    L.vd3d_depth_preprocess.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, vp]
    L.vd3d_gemm_x3_weight_bytes.argtypes = [i32, i32, i32]
    L.vd3d_gemm_x3_weight_bytes.restype = C.c_int64
    L.vd3d_gemm_x3_pack_weights.argtypes = [vp, vp, i32, i32, i32, vp]
    L.vd3d_gemm_x3.argtypes = [vp, vp, C.c_int64, i32, vp, i32, i32, vp, i32, vp]
    L.vd3d_attention_x3_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.vd3d_attention_x3_workspace_bytes.restype = C.c_int64
    L.vd3d_attention_x3.argtypes = [vp, vp, i32, i32, i32, i32, C.c_float, i32, vp, C.c_int64, vp]
    L.vd3d_conv3x3_x2_weight_bytes.argtypes = [i32, i32]
    L.vd3d_conv3x3_x2_weight_bytes.restype = C.c_int64
    L.vd3d_conv3x3_x2_pack_weights.argtypes = [vp, vp, i32, i32, vp]
    L.vd3d_conv3x3_x2.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp]
    L.vd3d_add_layernorm.argtypes = [vp, i32, vp, vp, vp, vp, C.c_float, C.c_int64, i32, vp, vp]
    L.vd3d_upsample_bilinear_nhwc.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32]
    L.vd3d_nhwc_bias_act_f32.argtypes = [vp, vp, vp, vp, vp, i32, C.c_int64, i32, vp, vp]
    L.vd3d_upsample_bilinear_bias_nhwc_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32]
    L.vd3d_dpt_head_tail_f32.argtypes = [vp, vp, vp, vp, C.c_float, C.c_float, C.c_int64, i32, vp]
    L.vd3d_preview_image.argtypes = [vp, i32, vp, vp, i32, i32, vp]
    L.vd3d_detect_black_bars.argtypes = [vp, vp, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.vd3d_stream_copy.argtypes = [vp, vp, vp, C.c_size_t]
    L.vd3d_torch_math.argtypes = [vp, i32, vp, C.c_float, vp, C.c_longlong]
    L.vd3d_torch_math_aten.argtypes = [vp, i32, vp, C.c_double, vp, C.c_longlong, i32]
    L.vd3d_debug_gaussian_kernel1d.argtypes = [i32, C.c_float, vp]
    L.vd3d_debug_exp_torch.argtypes = [vp, vp, C.c_longlong]
    L.vd3d_debug_tune.argtypes = [i32, i32]
    L.vd3d_set_profiling.argtypes = [vp, i32]
    L.vd3d_last_stage_ms.argtypes = [vp, C.c_char_p]
    L.vd3d_last_stage_ms.restype = C.c_float
    L.vd3d_stage_calls.argtypes = [vp, C.c_char_p]
    L.vd3d_stage_calls.restype = C.c_long
    L.vd3d_debug_planes.argtypes = [vp] + [C.POINTER(vp)] * 6
    L.vd3d_shift_params_default.argtypes = [C.POINTER(ShiftParams)]
    L.vd3d_render_params_default.argtypes = [C.POINTER(RenderParams)]
    if L.vd3d_abi_version() != _abi.ABI_VERSION:
        raise ImportError("libvd3d_hip.so ABI version mismatch")
    # development probes: VD3D_TUNE="knob:value,knob:value" applies vd3d_debug_tune at load time -- only with a development library (-DVD3D_DEV_KNOBS,
    # tools/build_ab.sh); the product library refuses the knobs and this loader says so instead of running an experiment that silently is not one
    if os.environ.get("VD3D_TUNE"):
        if L.vd3d_debug_tune(-1, 0) != 0:
            raise ImportError("VD3D_TUNE is set but this libvd3d_hip.so was built without -DVD3D_DEV_KNOBS (bash tools/build_ab.sh dev -DVD3D_DEV_KNOBS; VD3D_LIB_PATH=...)")
        for kv in filter(None, os.environ["VD3D_TUNE"].split(",")):
            k, v = kv.split(":")
            if L.vd3d_debug_tune(int(k), int(v)) != 0:
                raise ImportError(f"VD3D_TUNE: unknown knob {k}")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise Vd3dError(rc, lib().vd3d_last_error().decode(errors="replace"))
