"""Map the reference's Python call signatures onto the C-ABI parameter structs (pure host logic).

``render_kwargs_to_params`` accepts the keyword names of ``render_sbs_3d``
(core/render_3d.py:933-985) and reproduces which of them the reference actually forwards:
``depth_pop_*``, ``fg_pop_multiplier``, ``bg_push_multiplier``, ``subject_lock_strength`` are accepted
and IGNORED (the loop passes literals, :1299-1305), and ``parallax_balance`` is never forwarded
(:1284-1331), so the pixel_shift_cuda default 0.8 applies.
"""
from __future__ import annotations

import inspect
import os

from ._abi import RenderParams, ShiftParams
from .geometry import make_render_params, plan_geometry

# defaults of render_sbs_3d's keyword parameters (core/render_3d.py:949-984)
RENDER_DEFAULTS = dict(
    feather_strength=0.0, blur_ksize=1, use_ffmpeg=False, selected_ffmpeg_codec=None, crf_value=23,
    use_subject_tracking=False, use_floating_window=False, max_pixel_shift_percent=0.02, progress=None,
    progress_label=None, suspend_flag=None, cancel_flag=None, auto_crop_black_bars=False, parallax_balance=0.8,
    preserve_original_aspect=False, zero_parallax_strength=0.0, enable_edge_masking=True, enable_feathering=True,
    skip_blank_frames=False, original_video_width=None, original_video_height=None, convergence_strength=0.0,
    enable_dynamic_convergence=True, ipd_factor=1.0, depth_pop_gamma=0.85, depth_pop_mid=0.50, depth_stretch_lo=0.05,
    depth_stretch_hi=0.95, fg_pop_multiplier=1.20, bg_push_multiplier=1.10, subject_lock_strength=1.00,
    color_saturation=1.0, color_contrast=1.0, color_brightness=0.0, start_s=None, end_s=None,
)


def reference_aten_threads() -> int:
    """The N of the N-thread ATen mode for a DROP-IN caller (round 6): ``torch.get_num_threads()`` of the calling process.  The shims
    (``pixel_shift_cuda``, ``render_sbs_3d``, ``render_clip`` / ``render_pairs``) run INSIDE the reference's process, where that number is what
    the reference's own ``torch.mean`` (core/render_3d.py:418,928), ``torch.pow`` / ``torch.sigmoid`` (:209,517,620) and small-output bilinear
    ``F.interpolate`` (:595-596,1262-1263) would have run with -- so by default they reproduce the reference's thread-dependent float32
    arithmetic, not the thread-independent one.  ``VD3D_ATEN_THREADS`` overrides it (0 = the thread-independent arithmetic, the C-ABI default;
    N = a reference run on an N-thread machine).  Restated range: 1 .. 1024 threads.

    Assumption of the mode: the reference's torch is an x86-64 AVX-512 build on glibc >= 2.27 -- ATen's elementwise scalar
    tails are ``chunk_len mod 32`` elements (two 16-lane vectors per step; an AVX2-only host has ``mod 16``), ``expf`` is glibc's FMA ifunc
    variant.  On another ISA the mode is still deterministic but no longer the reference's bits at sizes with tails."""
    v = os.environ.get("VD3D_ATEN_THREADS")
    if v is not None and v.strip() != "":
        n = int(v)
    else:
        import torch
        n = int(torch.get_num_threads())
    if n < 0 or n > 1024:
        raise ValueError(f"ATen thread count {n} outside 0 .. 1024 (set VD3D_ATEN_THREADS)")
    return n


def shift_params_from_kwargs(fg_shift, mg_shift, bg_shift, **kw) -> ShiftParams:
    """pixel_shift_cuda(..., **kw) -> vd3d_shift_params (unknown keywords raise TypeError like Python would)."""
    kw = dict(kw)
    kw.pop("return_shift_map", None)
    kw.pop("dof_strength", None)  # accepted and unused by pixel_shift_cuda (:579)
    if kw.get("aten_threads") is None:   # extension keyword absent: the calling (reference) process's torch thread count -- reference_aten_threads()
        kw["aten_threads"] = reference_aten_threads()
    return ShiftParams.defaults(fg_shift, mg_shift, bg_shift, **kw)


def render_kwargs_to_params(src_w: int, src_h: int, *, output_height, fg_shift, mg_shift, bg_shift,
                            sharpness_factor, output_format, dof_strength, target_ratio=16 / 9, dof_dense_conv=True,
                            aten_sum_threads=0, **kw) -> RenderParams:
    """``dof_dense_conv`` (extension, not a render_sbs_3d parameter; default on): the DOF Gaussian levels in the reference's dense k x k
    association -- the mode that reproduces the reference's finishing stage bit for bit.  ``False`` selects the separable form
    (about 25 % less time in the finishing kernel, differs from the reference on ~0.5 % of samples; include/vd3d.h
    vd3d_render_params::dof_dense_conv).  ``aten_sum_threads`` (extension): N >= 1 reproduces the float32 ``torch.mean`` of the dynamic parallax scale
    and of the motion metric as torch computes them with N intra-op threads (``torch.get_num_threads()`` of the reference process); 0 = the correctly
    rounded exact mean (include/vd3d.h vd3d_render_params::aten_sum_threads).  This function keeps the C ABI's default (0); the drop-in entries
    (``render_pairs`` / ``render_clip`` / ``video_io.render_sbs_3d``) pass ``reference_aten_threads()`` unless told otherwise.  The mode assumes an
    AVX-512 / glibc reference host (see ``reference_aten_threads``)."""
    unknown = set(kw) - set(RENDER_DEFAULTS) - {"output_width", "input_path", "depth_path", "output_path",
                                                "selected_codec", "fps", "selected_aspect_ratio", "aspect_ratios"}
    if unknown:
        raise TypeError(f"render_sbs_3d() got unexpected keyword argument(s) {sorted(unknown)}")
    o = dict(RENDER_DEFAULTS)
    o.update(kw)
    # skip_blank_frames is loop-level (which frames are blank comes from ffmpeg's blackdetect, :1046-1060): render_clip /
    # Renderer.render_frame(blank=True) carry it, the per-clip parameter block does not change
    geom = plan_geometry(src_w, src_h, output_height, output_format, target_ratio, o["preserve_original_aspect"],
                         o["original_video_width"], o["original_video_height"])
    shift = ShiftParams.defaults(
        fg_shift, mg_shift, bg_shift,
        blur_ksize=o["blur_ksize"], feather_strength=o["feather_strength"],
        use_subject_tracking=o["use_subject_tracking"], enable_floating_window=o["use_floating_window"],
        max_pixel_shift_percent=o["max_pixel_shift_percent"], zero_parallax_strength=o["zero_parallax_strength"],
        enable_edge_masking=o["enable_edge_masking"], enable_feathering=o["enable_feathering"],
        convergence_strength=o["convergence_strength"], enable_dynamic_convergence=o["enable_dynamic_convergence"])
    p = make_render_params(geom, shift, ipd_factor=o["ipd_factor"], dof_strength=dof_strength,
                           sharpness_factor=sharpness_factor, color_saturation=o["color_saturation"],
                           color_contrast=o["color_contrast"], color_brightness=o["color_brightness"])
    p.auto_crop_black_bars = 1 if o["auto_crop_black_bars"] else 0   # :1230-1234, crop decided per frame on device
    p.target_ratio = float(target_ratio)
    p.dof_dense_conv = 1 if dof_dense_conv else 0
    p.aten_sum_threads = int(aten_sum_threads)
    return p
