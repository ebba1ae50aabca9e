"""Output geometry of the render loop: which sizes the loop of ``render_sbs_3d`` works at, as a rule table.

The reference derives, per clip (core/render_3d.py:1074-1138) and per frame (:1236-1259): the size the warp runs at, the
canvas each eye is fitted to, the writer size, the centre crop to the target aspect and the size both tensors are resized to
before the warp.  Here each output format is ONE row of ``_FORMAT_RULES``:

    eye canvas (fixed size, or derived from the warp size)   x   how many canvases wide the WRITER is opened
                                                             x   how many canvases wide the muxed frame really is

and ``plan_geometry`` evaluates the row.  Integer / float arithmetic is the reference's (``int()`` truncation, "make it even"
by adding 1, 0.01 aspect dead band).
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

# format -> (canvas rule, canvases per writer row, canvases per muxed row)
#   canvas rule: ("fixed", w, h) | ("warp",) the warp size | ("half",) half the warp width | ("hd_or_warp",) 1920x1080 unless the
#   original aspect is preserved (then the warp size)
# The writer of the interlaced / anaglyph formats is opened two canvases wide although one canvas is written
# (format_3d_output :851-858 vs :1134-1138): the muxed frame is what counts for the renderer.
_FORMAT_RULES = {
    "Full-SBS": (("hd_or_warp",), 2, 2),
    "Half-SBS": (("half",), 2, 2),
    "VR": (("fixed", 1440, 1600), 2, 2),
    "Red-Cyan Anaglyph": (("warp",), 2, 1),
    "Passive Interlaced": (("warp",), 2, 1),
}
_UNKNOWN_FORMAT = (("warp",), 2, 2)   # format_3d_output falls back to hstack for unknown strings (:860)


def _even(v: int) -> int:
    return v + (v & 1)


def plan_geometry(src_w: int, src_h: int, output_height: int, output_format: str = "Half-SBS",
                  target_ratio: float = 16 / 9, preserve_original_aspect: bool = False,
                  original_video_width: int | None = None, original_video_height: int | None = None) -> dict:
    """The integer geometry block of vd3d_render_params (+ the writer size) as a dict."""
    canvas_rule, writer_n, mux_n = _FORMAT_RULES.get(output_format, _UNKNOWN_FORMAT)
    fmt_id = FORMAT_IDS.get(output_format, FORMAT_IDS["Full-SBS"])

    # the size the warp runs at (:1086-1091 / :1110-1114)
    if preserve_original_aspect:
        warp_w = int(src_w if original_video_width is None or original_video_height is None else original_video_width)
        warp_h = int(src_h if original_video_width is None or original_video_height is None else original_video_height)
    else:
        warp_h = int(output_height)
        warp_w = _even(int(warp_h * target_ratio))

    # the canvas each eye is fitted to
    kind = canvas_rule[0]
    if kind == "fixed":
        fit_w, fit_h = canvas_rule[1], canvas_rule[2]
    elif kind == "half":
        fit_w, fit_h = warp_w // 2, warp_h
    elif kind == "hd_or_warp" and not preserve_original_aspect:
        fit_w, fit_h = 1920, 1080
    else:
        fit_w, fit_h = warp_w, warp_h
    # Half-SBS opens its writer at the warp size, not at two half canvases (odd widths differ by one)
    writer_w = warp_w if kind == "half" else fit_w * writer_n
    mux_w = fit_w * mux_n

    # centre crop of the decoded frame to the target aspect (:1236-1248): only beyond a 0.01 dead band
    crop = [0, 0, src_w, src_h]
    ratio = src_w / src_h
    if abs(ratio - target_ratio) > 0.01:
        if ratio > target_ratio:
            crop[2] = int(src_h * target_ratio)
            crop[0] = (src_w - crop[2]) // 2
        else:
            crop[3] = int(src_w / target_ratio)
            crop[1] = (src_h - crop[3]) // 2

    # the size both tensors are resized to before the warp (:1250-1259)
    eye_w, eye_h = fit_w, (fit_h if preserve_original_aspect else _even(int(fit_w / target_ratio)))

    return dict(src_w=src_w, src_h=src_h, crop_x=crop[0], crop_y=crop[1], crop_w=crop[2], crop_h=crop[3],
                eye_w=eye_w, eye_h=eye_h, warp_w=warp_w, warp_h=warp_h, fit_w=fit_w, fit_h=fit_h,
                out_w=mux_w, out_h=fit_h, format=fmt_id, writer_w=writer_w, writer_h=fit_h)


_GEOMETRY_FIELDS = ("src_w", "src_h", "crop_x", "crop_y", "crop_w", "crop_h", "eye_w", "eye_h", "warp_w", "warp_h",
                    "fit_w", "fit_h", "out_w", "out_h", "format")


def make_render_params(geom: dict, shift: ShiftParams, *, ipd_factor=1.0, dof_strength=2.0, sharpness_factor=0.2,
                       color_saturation=1.0, color_contrast=1.0, color_brightness=0.0) -> RenderParams:
    p = RenderParams()
    for k in _GEOMETRY_FIELDS:
        setattr(p, k, int(geom[k]))
    p.shift = shift
    for k, v in (("ipd_factor", ipd_factor), ("dof_strength", dof_strength), ("sharpness_factor", sharpness_factor),
                 ("color_saturation", color_saturation), ("color_contrast", color_contrast), ("color_brightness", color_brightness)):
        setattr(p, k, float(v))
    return p
