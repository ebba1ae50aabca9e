"""Preview visualisers -- host-side mirror of the reference's ``core/preview_utils.py:23-84`` over the HIP
library.  ``generate_preview_image`` keeps the reference's signature; the types whose definition is plain integer arithmetic
run as one HIP launch (``vd3d_preview_image``); the colour-mapped heat-maps compute their index plane on device
(``vd3d_preview_heatmap``) and look it up in OpenCV's own table, fetched from cv2 at first use or registered by the caller; the arrow
overlay (``vd3d_preview_arrows``) uses the closed forms OpenCV's rasteriser reduces to for horizontal arrows."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .render_3d import Renderer, _ptr, default_renderer

PREVIEW_TYPES = {"Passive Interlaced": 0, "HSBS": 1, "Left-Right Diff": 2, "Feather Blend": 3, "Red-Blue Anaglyph": 4}
# colour-mapped types: (vd3d_preview_heatmap type, OpenCV colour map)
HEATMAP_TYPES = {"Shift Heatmap": (0, "JET"), "Shift Heatmap (Abs)": (1, "JET"), "Shift Heatmap (Clipped \u00b15px)": (2, "JET"),
                 "Feather Mask": (3, "BONE")}
_COLORMAPS: dict = {}


def register_colormap(name: str, lut) -> None:
    """Install a 256 x 3 uint8 BGR table for ``name`` ("JET" / "BONE").  OpenCV's tables are that library's data: this package holds no
    copy, it asks cv2 for them on first use (``colormap_lut``) or takes what the caller registers here."""
    lut = np.ascontiguousarray(lut, np.uint8).reshape(256, 3)
    _COLORMAPS[name.upper()] = lut


def colormap_lut(name: str) -> np.ndarray:
    key = name.upper()
    if key not in _COLORMAPS:
        try:
            import cv2
        except ImportError as e:
            raise NotImplementedError(f"colour map {name!r}: OpenCV is not importable and no table was registered "
                                      "(preview_utils.register_colormap)") from e
        ramp = np.arange(256, dtype=np.uint8).reshape(256, 1)
        _COLORMAPS[key] = cv2.applyColorMap(ramp, getattr(cv2, "COLORMAP_" + key)).reshape(256, 3).copy()
    return _COLORMAPS[key]


def preview_heatmap(renderer: Renderer, preview_type: str, shift_map: torch.Tensor, lut=None) -> torch.Tensor:
    """Device tensor in (float32 shift map [1,h,w] or [h,w]), uint8 BGR [h,w,3] device tensor out."""
    kind, cmap = HEATMAP_TYPES[preview_type]
    s = shift_map.to(renderer.device, torch.float32)
    if s.dim() == 3 and s.shape[0] == 1:
        s = s[0]
    if s.dim() != 2:
        raise AssertionError("shift_map must be [h,w] or [1,h,w]")
    s = s.contiguous()
    table = torch.from_numpy(colormap_lut(cmap) if lut is None else np.ascontiguousarray(lut, np.uint8).reshape(256, 3)).to(renderer.device)
    h, w = int(s.shape[0]), int(s.shape[1])
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=renderer.device)
    renderer._enter(s, table, out)
    _lib.check(renderer._L.vd3d_preview_heatmap(renderer._ctx, kind, _ptr(s), h, w, _ptr(table), _ptr(out)))
    return out


def preview_arrows(renderer: Renderer, left: torch.Tensor, shift_map: torch.Tensor) -> torch.Tensor:
    """ "Overlay Arrows" (core/preview_utils.py:74-82): uint8 BGR [h,w,3] left eye + float32 shift map -> the eye with green arrows."""
    l_ = left.to(renderer.device, torch.uint8).contiguous()
    s = shift_map.to(renderer.device, torch.float32)
    if s.dim() == 3 and s.shape[0] == 1:
        s = s[0]
    s = s.contiguous()
    if l_.dim() != 3 or l_.shape[2] != 3 or tuple(s.shape) != tuple(l_.shape[:2]):
        raise AssertionError("left must be uint8 [h,w,3] and shift_map [h,w] / [1,h,w] of the same size")
    out = torch.empty_like(l_)
    renderer._enter(l_, s, out)
    _lib.check(renderer._L.vd3d_preview_arrows(renderer._ctx, _ptr(l_), _ptr(s), int(l_.shape[0]), int(l_.shape[1]), _ptr(out)))
    return out


def preview_image(renderer: Renderer, preview_type: str, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
    """Device tensors in (uint8 BGR [h,w,3] eyes), device tensor out."""
    if preview_type not in PREVIEW_TYPES:
        raise NotImplementedError(f"preview type {preview_type!r} is not an eye-only preview (heat-maps: preview_heatmap; arrows: preview_arrows)")
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
    """Reference signature (core/preview_utils.py:23): NumPy BGR eyes + the shift-map tensor in, NumPy BGR preview out (None for unknown
    types)."""
    r = default_renderer()
    if preview_type == "Overlay Arrows":
        sm = shift_map if torch.is_tensor(shift_map) else torch.from_numpy(np.asarray(shift_map))
        H, W = int(left.shape[0]), int(left.shape[1])
        SH_, SW_ = int(sm.shape[-2]), int(sm.shape[-1])
        # the reference indexes shift_np[y, x] at the grid points range(0, h, 20) x range(0, w, 20) only (:77-79): the LAST grid point has
        # to exist in the shift map, (w, h) itself may exceed it
        if int(h) > 0 and int(w) > 0 and (((int(h) - 1) // 20) * 20 >= SH_ or ((int(w) - 1) // 20) * 20 >= SW_):
            raise IndexError("Overlay Arrows: a grid point of (w, h) lies outside the shift map")
        if (int(h), int(w)) != (H, W):   # arrows start only inside the (w, h) grid: a zero shift draws nothing (|dx| <= 1)
            sm = sm.clone()
            sm[..., int(h):, :] = 0
            sm[..., :, int(w):] = 0
        return preview_arrows(r, torch.from_numpy(np.ascontiguousarray(left)), sm).cpu().numpy()
    if preview_type in HEATMAP_TYPES:
        return preview_heatmap(r, preview_type, shift_map if torch.is_tensor(shift_map) else torch.from_numpy(np.asarray(shift_map))).cpu().numpy()
    if preview_type not in PREVIEW_TYPES:
        return None
    lt = torch.from_numpy(np.ascontiguousarray(left))
    rt = torch.from_numpy(np.ascontiguousarray(right))
    return preview_image(r, preview_type, lt, rt).cpu().numpy()
