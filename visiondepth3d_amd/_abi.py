"""ctypes mirror of include/vd3d.h (struct layouts + enums).  Pure data definitions."""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 6

FMT_HALF_SBS, FMT_FULL_SBS, FMT_VR, FMT_ANAGLYPH, FMT_INTERLACED = range(5)
FORMAT_IDS = {
    "Half-SBS": FMT_HALF_SBS,
    "Full-SBS": FMT_FULL_SBS,
    "VR": FMT_VR,
    "Red-Cyan Anaglyph": FMT_ANAGLYPH,
    "Passive Interlaced": FMT_INTERLACED,
}
DEPTH_F32, DEPTH_BGR_U8, DEPTH_GRAY_U8 = range(3)
DT_BF16, DT_F32, DT_F16 = range(3)   # vd3d_dtype (DT_F16: vd3d_esr_preprocess only)

E_INVALID, E_HIP, E_NOMEM, E_UNSUPPORTED = -1, -2, -3, -4


class ShiftParams(C.Structure):
    """vd3d_shift_params: keyword parameters of pixel_shift_cuda (core/render_3d.py:561-590)."""
    _fields_ = [
        ("fg_shift", C.c_double), ("mg_shift", C.c_double), ("bg_shift", C.c_double),
        ("blur_ksize", C.c_int32),
        ("use_subject_tracking", C.c_int32),
        ("enable_floating_window", C.c_int32),
        ("enable_feathering", C.c_int32),
        ("enable_edge_masking", C.c_int32),
        ("enable_dynamic_convergence", C.c_int32),
        ("feather_strength", C.c_double),
        ("max_pixel_shift_percent", C.c_double),
        ("parallax_balance", C.c_double),
        ("zero_parallax_strength", C.c_double),
        ("convergence_strength", C.c_double),
        ("depth_pop_gamma", C.c_double),
        ("depth_pop_mid", C.c_double),
        ("depth_stretch_lo", C.c_double),
        ("depth_stretch_hi", C.c_double),
        ("fg_pop_multiplier", C.c_double),
        ("bg_push_multiplier", C.c_double),
        ("subject_lock_strength", C.c_double),
        ("aten_threads", C.c_int32),    # extension: ATen's scalar tails for a reference running N torch threads (include/vd3d.h); 0 = none
        ("reserved0", C.c_int32),
    ]

    @classmethod
    def defaults(cls, fg_shift=0.0, mg_shift=0.0, bg_shift=0.0, **kw) -> "ShiftParams":
        p = cls(fg_shift=float(fg_shift), mg_shift=float(mg_shift), bg_shift=float(bg_shift),
                blur_ksize=9, use_subject_tracking=1, enable_floating_window=1, enable_feathering=1,
                enable_edge_masking=1, enable_dynamic_convergence=1, feather_strength=10.0,
                max_pixel_shift_percent=0.02, parallax_balance=0.8, zero_parallax_strength=0.0,
                convergence_strength=0.0, depth_pop_gamma=0.85, depth_pop_mid=0.50, depth_stretch_lo=0.05,
                depth_stretch_hi=0.95, fg_pop_multiplier=1.20, bg_push_multiplier=1.10,
                subject_lock_strength=1.00, aten_threads=0, reserved0=0)
        for k, v in kw.items():
            if k not in dict(cls._fields_):
                raise TypeError(f"unknown pixel_shift_cuda parameter {k!r}")
            setattr(p, k, int(v) if dict(cls._fields_)[k] is C.c_int32 else float(v))
        return p


class RenderParams(C.Structure):
    """vd3d_render_params: per-clip constants of the render_sbs_3d loop body (core/render_3d.py:1227-1419)."""
    _fields_ = [
        ("src_w", C.c_int32), ("src_h", C.c_int32),
        ("crop_x", C.c_int32), ("crop_y", C.c_int32), ("crop_w", C.c_int32), ("crop_h", C.c_int32),
        ("eye_w", C.c_int32), ("eye_h", C.c_int32),
        ("warp_w", C.c_int32), ("warp_h", C.c_int32),
        ("fit_w", C.c_int32), ("fit_h", C.c_int32),
        ("out_w", C.c_int32), ("out_h", C.c_int32),
        ("format", C.c_int32), ("auto_crop_black_bars", C.c_int32),
        ("shift", ShiftParams),
        ("ipd_factor", C.c_double),
        ("dof_strength", C.c_double),
        ("sharpness_factor", C.c_double),
        ("color_saturation", C.c_double), ("color_contrast", C.c_double), ("color_brightness", C.c_double),
        ("target_ratio", C.c_double),
        ("dof_dense_conv", C.c_int32), ("aten_sum_threads", C.c_int32),
    ]


class State(C.Structure):
    """vd3d_state: every temporal-tracker scalar of the reference made explicit."""
    _fields_ = [
        ("fw_prev_offset", C.c_double), ("fw_frame_counter", C.c_int32),
        ("ema_valid", C.c_int32), ("ema_lo", C.c_float), ("ema_hi", C.c_float),
        ("conv_valid", C.c_int32), ("bar_prev_width", C.c_int32), ("conv_val", C.c_double),
        ("focal_valid", C.c_int32), ("smooth_valid", C.c_int32), ("focal", C.c_double),
        ("sm_fg", C.c_double), ("sm_mg", C.c_double), ("sm_bg", C.c_double),
        ("tdf_valid", C.c_int32), ("prev_depth_valid", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class FrameScalars(C.Structure):
    """vd3d_frame_scalars: per-frame scalars produced by the device-side scalar stage."""
    _fields_ = [
        ("q_lo", C.c_float), ("q_hi", C.c_float), ("ema_lo", C.c_float), ("ema_hi", C.c_float),
        ("mean_c", C.c_float), ("var_c", C.c_float), ("dyn_scale", C.c_double),
        ("s_norm", C.c_float), ("mad", C.c_float),
        ("s0", C.c_float), ("q05", C.c_float), ("q95", C.c_float), ("s1", C.c_float),
        ("zpo_raw", C.c_float), ("zpo", C.c_double), ("focal", C.c_double), ("stable_zero", C.c_double),
        ("bar_width", C.c_int32), ("bar_side", C.c_int32), ("collapse", C.c_int32),
        ("crop_top", C.c_int32), ("crop_bottom", C.c_int32), ("reserved", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
