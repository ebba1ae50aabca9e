"""Output geometry of the render loop -- host-side mirror of core/render_3d.py:1074-1138 (per clip)
and :1236-1259 (per frame).  Pure integer/float Python, identical arithmetic to the reference.
"""
from __future__ import annotations

from ._abi import FORMAT_IDS, RenderParams, ShiftParams

# render_3d's own 7-entry table (core/render_3d.py:39-47); the GUI passes its 14-entry dict instead
aspect_ratios = {
    "Default (16:9)": 16 / 9,
    "CinemaScope (2.39:1)": 2.39,
    "21:9 UltraWide": 21 / 9,
    "4:3 (Classic Films)": 4 / 3,
    "1:1 (Square)": 1 / 1,
    "2.35:1 (Classic Cinematic)": 2.35,
    "2.76:1 (Ultra-Panavision)": 2.76,
}


def plan_geometry(src_w: int, src_h: int, output_height: int, output_format: str = "Half-SBS",
                  target_ratio: float = 16 / 9, preserve_original_aspect: bool = False,
                  original_video_width: int | None = None, original_video_height: int | None = None) -> dict:
    """Return the integer geometry block of vd3d_render_params as a dict."""
    if output_format not in FORMAT_IDS:
        # format_3d_output falls back to hstack for unknown strings (:860); geometry takes the 'else' branch
        fmt_id = FORMAT_IDS["Full-SBS"]
    else:
        fmt_id = FORMAT_IDS[output_format]

    # --- per-clip (:1086-1138)
    if preserve_original_aspect:
        if original_video_width is None or original_video_height is None:
            original_video_width, original_video_height = src_w, src_h
        resized_width, resized_height = int(original_video_width), int(original_video_height)
        if output_format == "Full-SBS":
            per_eye_w, per_eye_h = resized_width, resized_height
            out_width, out_height = per_eye_w * 2, per_eye_h
        elif output_format == "Half-SBS":
            per_eye_w, per_eye_h = resized_width // 2, resized_height
            out_width, out_height = resized_width, resized_height
        elif output_format == "VR":
            per_eye_w, per_eye_h = 1440, 1600
            out_width, out_height = per_eye_w * 2, per_eye_h
        else:
            per_eye_w, per_eye_h = resized_width, resized_height
            out_width, out_height = resized_width * 2, resized_height
    else:
        resized_height = int(output_height)
        resized_width = int(resized_height * target_ratio)
        if resized_width % 2 != 0:
            resized_width += 1
        if output_format == "Full-SBS":
            per_eye_w, per_eye_h = 1920, 1080
            out_width, out_height = per_eye_w * 2, per_eye_h
        elif output_format == "Half-SBS":
            per_eye_w, per_eye_h = resized_width // 2, resized_height
            out_width, out_height = resized_width, resized_height
        elif output_format == "VR":
            per_eye_w, per_eye_h = 1440, 1600
            out_width, out_height = per_eye_w * 2, per_eye_h
        else:
            per_eye_w, per_eye_h = resized_width, resized_height
            out_width, out_height = resized_width * 2, resized_height

    # --- per-frame centre crop (:1236-1248)
    h, w = src_h, src_w
    crop_x = crop_y = 0
    crop_w, crop_h = w, h
    current_ratio = w / h
    if abs(current_ratio - target_ratio) > 0.01:
        if current_ratio > target_ratio:
            new_w = int(h * target_ratio)
            crop_x = (w - new_w) // 2
            crop_w = new_w
        else:
            new_h = int(w / target_ratio)
            crop_y = (h - new_h) // 2
            crop_h = new_h

    # --- per-eye size (:1250-1259)
    if not preserve_original_aspect:
        target_eye_w = per_eye_w
        target_eye_h = int(per_eye_w / target_ratio)
        if target_eye_h % 2 != 0:
            target_eye_h += 1
    else:
        target_eye_w, target_eye_h = per_eye_w, per_eye_h

    # Passive Interlaced / Anaglyph write one eye-sized frame although the writer was opened 2x wide
    # (format_3d_output :851-858 vs out_width :1134-1138): the muxed frame is fit_w x fit_h.
    if fmt_id in (FORMAT_IDS["Red-Cyan Anaglyph"], FORMAT_IDS["Passive Interlaced"]):
        mux_w, mux_h = per_eye_w, per_eye_h
    else:
        mux_w, mux_h = per_eye_w * 2, per_eye_h

    return dict(src_w=src_w, src_h=src_h, crop_x=crop_x, crop_y=crop_y, crop_w=crop_w, crop_h=crop_h,
                eye_w=target_eye_w, eye_h=target_eye_h, warp_w=resized_width, warp_h=resized_height,
                fit_w=per_eye_w, fit_h=per_eye_h, out_w=mux_w, out_h=mux_h, format=fmt_id,
                writer_w=out_width, writer_h=out_height)


def make_render_params(geom: dict, shift: ShiftParams, *, ipd_factor=1.0, dof_strength=2.0, sharpness_factor=0.2,
                       color_saturation=1.0, color_contrast=1.0, color_brightness=0.0) -> RenderParams:
    p = RenderParams()
    for k in ("src_w", "src_h", "crop_x", "crop_y", "crop_w", "crop_h", "eye_w", "eye_h", "warp_w", "warp_h",
              "fit_w", "fit_h", "out_w", "out_h", "format"):
        setattr(p, k, int(geom[k]))
    p.shift = shift
    p.ipd_factor = float(ipd_factor)
    p.dof_strength = float(dof_strength)
    p.sharpness_factor = float(sharpness_factor)
    p.color_saturation = float(color_saturation)
    p.color_contrast = float(color_contrast)
    p.color_brightness = float(color_brightness)
    return p
