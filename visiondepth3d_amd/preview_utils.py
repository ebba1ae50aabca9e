"""Preview visualisers -- host-side mirror of the reference's ``core/preview_utils.py:23-84`` over the HIP
library.  ``generate_preview_image`` keeps the reference's signature; the types whose definition is plain integer arithmetic
run as one HIP launch (``vd3d_preview_image``), the colour-mapped heat-maps and the arrow overlay (OpenCV colour-map tables
and line rasteriser) raise ``NotImplementedError`` -- never an approximation."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .render_3d import Renderer, _ptr, default_renderer

PREVIEW_TYPES = {"Passive Interlaced": 0, "HSBS": 1, "Left-Right Diff": 2, "Feather Blend": 3, "Red-Blue Anaglyph": 4}
UNSUPPORTED = ("Shift Heatmap", "Shift Heatmap (Abs)", "Shift Heatmap (Clipped \u00b15px)", "Feather Mask", "Overlay Arrows")


def preview_image(renderer: Renderer, preview_type: str, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
    """Device tensors in (uint8 BGR [h,w,3] eyes), device tensor out."""
    if preview_type not in PREVIEW_TYPES:
        raise NotImplementedError(f"preview type {preview_type!r} needs OpenCV's colour-map tables / line rasteriser (not built)")
    l_ = left.to(renderer.device, torch.uint8).contiguous()
    r_ = right.to(renderer.device, torch.uint8).contiguous()
    if l_.shape != r_.shape or l_.dim() != 3 or l_.shape[2] != 3:
        raise AssertionError("left / right must be uint8 [h,w,3] frames of the same size")
    h, w = int(l_.shape[0]), int(l_.shape[1])
    t = PREVIEW_TYPES[preview_type]
    out = torch.empty((h, 2 * (w // 2) if t == 1 else w, 3), dtype=torch.uint8, device=renderer.device)
    renderer._enter(l_, r_, out)
    _lib.check(renderer._L.vd3d_preview_image(renderer._ctx, t, _ptr(l_), _ptr(r_), h, w, _ptr(out)))
    return out


def generate_preview_image(preview_type, left, right, shift_map, w, h):
    """Reference signature (core/preview_utils.py:23): NumPy BGR eyes in, NumPy BGR preview out (None for unknown types)."""
    if preview_type in UNSUPPORTED:
        raise NotImplementedError(f"preview type {preview_type!r} needs OpenCV's colour-map tables / line rasteriser (not built)")
    if preview_type not in PREVIEW_TYPES:
        return None
    r = default_renderer()
    lt = torch.from_numpy(np.ascontiguousarray(left))
    rt = torch.from_numpy(np.ascontiguousarray(right))
    return preview_image(r, preview_type, lt, rt).cpu().numpy()
