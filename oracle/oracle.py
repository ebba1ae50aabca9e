"""ctypes front-end of the CPU oracle (oracle/vd3d_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from visiondepth3d_amd._abi import FrameScalars, RenderParams, ShiftParams, State

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvd3d_oracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


class ShiftDebug(C.Structure):
    _fields_ = [("s0", C.c_float), ("q05", C.c_float), ("q95", C.c_float), ("s1", C.c_float),
                ("zpo_raw", C.c_float), ("zpo", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vd3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.vo_quantile.restype = C.c_float
        _lib.vo_quantile.argtypes = [_f32p, C.c_int64, C.c_float]
        _lib.vo_subject_depth.restype = C.c_float
        _lib.vo_dynamic_parallax_scale.restype = C.c_double
        _lib.vo_dynamic_parallax_scale.argtypes = [_f32p, C.c_int, C.c_int, C.c_double, C.c_double, _f32p, _f32p]
        _lib.vo_motion_metric.restype = C.c_double
        _lib.vo_motion_metric.argtypes = [_f32p, _f32p, C.c_size_t, _f32p]
        _lib.vo_fw_smooth_offset.restype = C.c_double
        _lib.vo_fw_smooth_offset.argtypes = [C.POINTER(State), C.c_double, C.c_double]
        _lib.vo_focal_update.restype = C.c_double
        _lib.vo_focal_update.argtypes = [C.POINTER(State), C.c_double, C.c_double]
        _lib.vo_conv_update.restype = C.c_double
        _lib.vo_conv_update.argtypes = [C.POINTER(State), C.c_double]
        _lib.vo_bar_ease.argtypes = [C.POINTER(State), C.c_int]
        _lib.vo_apply_dof.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_double, C.c_double, _f32p]
        _lib.vo_color_grade.argtypes = [_f32p, C.c_size_t, C.c_double, C.c_double, C.c_double, _f32p]
        _lib.vo_sharpen.argtypes = [_u8p, C.c_int, C.c_int, C.c_double, _u8p]
        _lib.vo_shape_depth_for_pop.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float,
                                                C.c_float, _f32p, _f32p, _f32p]
        _lib.vo_shape_depth_for_pop_aten.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float,
                                                     C.c_double, C.c_int, _f32p, _f32p, _f32p]
        _lib.vo_interp_bilinear_aten.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int]
        _lib.vo_torch_math_aten.argtypes = [C.c_int, _f32p, C.c_double, _f32p, C.c_longlong, C.c_int]
        _lib.vo_expf_glibc.restype = C.c_float
        _lib.vo_expf_glibc.argtypes = [C.c_float]
        _lib.vo_curvature_clamp.argtypes = [_f32p, C.c_int, C.c_int, C.c_float]
        _lib.vo_gaussian_kernel1d.argtypes = [C.c_int, C.c_float, _f32p]
        _lib.vo_finish_frame.argtypes = [_u8p, _u8p, _f32p, C.c_int, C.c_int, C.POINTER(RenderParams), C.c_double,
                                         C.c_int, C.c_int, _u8p]
        _lib.vo_temporal_filter.argtypes = [C.POINTER(State), _f32p, _f32p, C.c_size_t]
        _lib.vo_percentile_ema_normalize.argtypes = [C.POINTER(State), _f32p, C.c_size_t, _f32p, _f32p, _f32p]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _u(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


# ----------------------------------------------------------------------------- leaf functions
def frame_to_tensor(bgr):
    bgr, pb = _u(bgr)
    h, w = bgr.shape[:2]
    out = np.empty((3, h, w), np.float32)
    lib().vo_frame_to_tensor(pb, h, w, out.ctypes.data_as(_f32p))
    return out


def depth_to_tensor(bgr):
    bgr, pb = _u(bgr)
    h, w = bgr.shape[:2]
    out = np.empty((1, h, w), np.float32)
    lib().vo_depth_to_tensor(pb, h, w, out.ctypes.data_as(_f32p))
    return out


def tensor_to_frame(t):
    t, pt = _f(t)
    _, h, w = t.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().vo_tensor_to_frame(pt, h, w, out.ctypes.data_as(_u8p))
    return out


def interp_bilinear(t, oh, ow, aten_threads=0):
    """F.interpolate(bilinear, align_corners=False).  ``aten_threads`` >= 1: the kernel ATen picks for a process with that many intra-op threads (its
    premultiplied channels_last kernel for outputs with oh + ow <= 128, and for 3-channel inputs with one thread); 0: the nested form at every size."""
    t, pt = _f(t)
    c, ih, iw = t.shape
    out = np.empty((c, oh, ow), np.float32)
    lib().vo_interp_bilinear_aten(pt, c, ih, iw, out.ctypes.data_as(_f32p), oh, ow, int(aten_threads))
    return out


def torch_math_aten(op, x, param=0.0, aten_threads=0):
    """torch.pow(x, param) (op 0) / torch.sigmoid(x) (op 1) of a contiguous float32 tensor as a torch process with ``aten_threads`` intra-op threads
    computes them: SLEEF vector bodies, libm on the scalar tails of every thread's chunk (vd3d_oracle.c, "ATen's SCALAR TAILS")."""
    x, px = _f(x)
    out = np.empty_like(x)
    lib().vo_torch_math_aten(int(op), px, float(param), out.ctypes.data_as(_f32p), x.size, int(aten_threads))
    return out


def expf_glibc(x):
    return float(lib().vo_expf_glibc(C.c_float(np.float32(x))))


def torch_math(op, x, param=0.0):
    """torch.pow(x, param) / torch.sigmoid(x) / torch.sqrt(x) as torch's CPU kernels evaluate them (pow_torch / sigmoid_torch /
    sqrt_torch of vd3d_oracle.c), elementwise."""
    x, px = _f(x)
    out = np.empty_like(x)
    lib().vo_torch_math(C.c_int({"pow": 0, "sigmoid": 1, "sqrt": 2, "exp": 3}[op]), px, C.c_float(np.float32(param)), out.ctypes.data_as(_f32p),
                        C.c_longlong(x.size))
    return out


def avg_pool2d(plane, k):
    """F.avg_pool2d(plane[None], k, stride=1, padding=k // 2) on one [H, W] plane in ATen's summation order."""
    a, pa = _f(plane)
    H, W = a.shape
    out = np.empty_like(a)
    lib().vo_avg_pool2d(pa, C.c_int(H), C.c_int(W), C.c_int(int(k)), out.ctypes.data_as(_f32p))
    return out


def grid_sample(plane, grid_xy):
    """F.grid_sample(plane[None, None], grid[None], mode='bilinear', padding_mode='border', align_corners=True) for one [H, W] plane and a
    [H, W, 2] grid of the same size (the reference's use, core/render_3d.py:697-701)."""
    a, pa = _f(plane)
    g, pg = _f(grid_xy)
    H, W = a.shape
    assert g.shape == (H, W, 2)
    out = np.empty_like(a)
    lib().vo_grid_sample(pa, C.c_int(H), C.c_int(W), pg, out.ctypes.data_as(_f32p))
    return out


def gaussian_blur_dense(plane, k, sigma):
    """torchvision.transforms.functional.gaussian_blur(plane, k, sigma) on one [H, W] plane in the dense k x k order of PyTorch's CPU build."""
    a, pa = _f(plane)
    H, W = a.shape
    out = np.empty_like(a)
    lib().vo_gaussian_blur_dense(pa, C.c_int(H), C.c_int(W), C.c_int(int(k)), C.c_float(np.float32(sigma)), out.ctypes.data_as(_f32p))
    return out


def sum_aten(v):
    """torch.sum of a contiguous float32 vector in ATen's CPU order (sum_aten_f32)."""
    a, pa = _f(np.ravel(v))
    L = lib()
    L.vo_sum_aten.restype = C.c_float
    return np.float32(L.vo_sum_aten(pa, C.c_int(a.size)))


def linspace(start, end, steps):
    L = lib()
    L.vo_linspace.restype = C.c_float
    return np.array([L.vo_linspace(C.c_float(start), C.c_float(end), C.c_int(steps), C.c_int(i)) for i in range(steps)], np.float32)


def quantile(v, q):
    v, pv = _f(np.ravel(v))
    return float(lib().vo_quantile(pv, v.size, C.c_float(np.float32(q))))


def subject_depth(d):
    d, pd = _f(d)
    H, W = d.shape[-2:]
    return float(lib().vo_subject_depth(pd, H, W))


def dynamic_parallax_scale(d, min_scale=0.6, max_scale=1.0, aten_threads=0):
    """aten_threads > 0: torch.mean of the centre crop as torch computes it with that many intra-op threads (ATen's float32 cascade sum)."""
    d, pd = _f(d)
    H, W = d.shape[-2:]
    m, v = C.c_float(), C.c_float()
    lib().vo_set_aten_threads(int(aten_threads))
    try:
        return float(lib().vo_dynamic_parallax_scale(pd, H, W, min_scale, max_scale, C.byref(m), C.byref(v)))
    finally:
        lib().vo_set_aten_threads(0)


def motion_metric(prev, cur, aten_threads=0):
    prev, pp = _f(prev)
    cur, pc = _f(cur)
    mad = C.c_float()
    lib().vo_set_aten_threads(int(aten_threads))
    try:
        return float(lib().vo_motion_metric(pp, pc, cur.size, C.byref(mad)))
    finally:
        lib().vo_set_aten_threads(0)


def sum_aten_2d(view, threads=1):
    """torch.sum of a float32 2-D view (rows may be strided, columns contiguous) as ATen's CPU kernel adds it with `threads` intra-op threads."""
    v = np.asarray(view)
    assert v.dtype == np.float32 and v.ndim == 2 and v.strides[1] == 4 and v.strides[0] % 4 == 0
    L = lib()
    L.vo_sum_aten_2d.restype = C.c_float
    L.vo_sum_aten_2d.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int]
    return np.float32(L.vo_sum_aten_2d(v.ctypes.data, v.shape[0], v.shape[1], v.strides[0] // 4, int(threads)))


def curvature_clamp(d, strength=0.08):
    d = np.array(d, dtype=np.float32, copy=True)
    H, W = d.shape[-2:]
    lib().vo_curvature_clamp(d.ctypes.data_as(_f32p), H, W, C.c_float(np.float32(strength)))
    return d


def shape_depth_for_pop(d, subject, stretch_lo=0.05, stretch_hi=0.95, depth_mid=0.5, gamma=0.85, aten_threads=0):
    d, pd = _f(d)
    out = np.empty_like(d)
    lo, hi = C.c_float(), C.c_float()
    f = lambda x: C.c_float(np.float32(x))
    lib().vo_shape_depth_for_pop_aten(pd, d.size, f(subject), f(stretch_lo), f(stretch_hi), f(depth_mid), float(gamma), int(aten_threads),
                                      out.ctypes.data_as(_f32p), C.byref(lo), C.byref(hi))
    return out, lo.value, hi.value


def heal_missing_pixels(warped, orig, edge_mask=None, heal_strength=0.5):
    """heal_missing_pixels(warped_frame, warped_depth, original_frame, edge_mask, heal_strength) (core/render_3d.py:431-459)."""
    w, pw = _f(warped)
    o, po = _f(orig)
    _, H, W = w.shape
    out = np.empty_like(w)
    pe = None
    if edge_mask is not None:
        e, pe = _f(np.reshape(edge_mask, (H, W)))
    lib().vo_heal_missing_pixels(pw, po, pe, H, W, C.c_double(float(heal_strength)), out.ctypes.data_as(_f32p))
    return out


def gaussian_kernel1d(k, sigma):
    out = np.empty(k, np.float32)
    lib().vo_gaussian_kernel1d(k, C.c_float(np.float32(sigma)), out.ctypes.data_as(_f32p))
    return out


def apply_dof(rgb, depth, focal_depth, max_sigma=2.0):
    rgb, pr = _f(rgb)
    depth, pd = _f(depth)
    _, H, W = rgb.shape
    out = np.empty_like(rgb)
    lib().vo_apply_dof(pr, pd, H, W, float(focal_depth), float(max_sigma), out.ctypes.data_as(_f32p))
    return out


def color_grade(rgb, saturation=1.0, contrast=1.0, brightness=0.0):
    rgb, pr = _f(rgb)
    out = np.empty_like(rgb)
    lib().vo_color_grade(pr, rgb.shape[1] * rgb.shape[2], float(saturation), float(contrast), float(brightness),
                         out.ctypes.data_as(_f32p))
    return out


def sharpen(img, factor=1.0):
    img, pi = _u(img)
    out = np.empty_like(img)
    lib().vo_sharpen(pi, img.shape[0], img.shape[1], float(factor), out.ctypes.data_as(_u8p))
    return out


def side_mask(img, side, width):
    img = np.array(img, dtype=np.uint8, copy=True)
    lib().vo_side_mask(img.ctypes.data_as(_u8p), img.shape[0], img.shape[1],
                       {"right": 1, "left": 2, None: 0}.get(side, side), int(width))
    return img


def resize_area_int(img, dw, dh):
    img, pi = _u(img)
    out = np.empty((dh, dw, 3), np.uint8)
    rc = lib().vo_resize_area_int(pi, img.shape[0], img.shape[1], out.ctypes.data_as(_u8p), dh, dw)
    if rc:
        raise NotImplementedError("oracle: non-integer INTER_AREA ratio")
    return out


def resize_area(img, dw, dh):
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_AREA), any down-scale ratio."""
    img, pi = _u(img)
    out = np.empty((dh, dw, 3), np.uint8)
    rc = lib().vo_resize_area(pi, img.shape[0], img.shape[1], out.ctypes.data_as(_u8p), dh, dw)
    if rc:
        raise NotImplementedError("oracle: INTER_AREA up-scale")
    return out


PREVIEW_TYPES = {"Passive Interlaced": 0, "HSBS": 1, "Left-Right Diff": 2, "Feather Blend": 3, "Red-Blue Anaglyph": 4}


def preview_image(preview_type, left, right):
    """generate_preview_image(preview_type, left, right, shift_map, w, h) for the exactly defined types."""
    if preview_type not in PREVIEW_TYPES:
        raise NotImplementedError(f"oracle: preview type {preview_type!r}")
    L, pl = _u(left)
    R, pr = _u(right)
    h, w = L.shape[:2]
    t = PREVIEW_TYPES[preview_type]
    out = np.empty((h, 2 * (w // 2) if t == 1 else w, 3), np.uint8)
    if lib().vo_preview_image(t, pl, pr, h, w, out.ctypes.data_as(_u8p)):
        raise NotImplementedError(f"oracle: preview type {preview_type!r}")
    return out


def detect_black_bars(frame_bgr):
    """detect_black_bars(frame_to_tensor(frame_bgr)) -> (top, bottom)."""
    img, pi = _u(frame_bgr)
    t, b = C.c_int(), C.c_int()
    lib().vo_detect_black_bars(pi, img.shape[0], img.shape[1], C.byref(t), C.byref(b))
    return t.value, b.value


def pad_to_aspect(img, tw, th):
    img, pi = _u(img)
    out = np.empty((th, tw, 3), np.uint8)
    rc = lib().vo_pad_to_aspect(pi, img.shape[0], img.shape[1], out.ctypes.data_as(_u8p), th, tw)
    if rc:
        raise NotImplementedError("oracle: non-integer INTER_AREA ratio")
    return out


def format_output(L, R, fmt):
    L, pl = _u(L)
    R, pr = _u(R)
    h, w = L.shape[:2]
    out = np.empty((1600, 2880, 3) if fmt == 2 else ((h, 2 * w, 3) if fmt in (0, 1) else (h, w, 3)), np.uint8)
    rc = lib().vo_format_output(pl, pr, h, w, int(fmt), out.ctypes.data_as(_u8p))
    if rc:
        raise NotImplementedError(f"oracle: format {fmt}")
    return out


def depth_handoff(pred, H, W, invert=False):
    """a24: bicubic post-process to (H, W) + per-frame min-max -> uint8 (truncation)."""
    pred, pp = _f(pred)
    ph, pw = pred.shape[-2:]
    out = np.empty((H, W), np.uint8)
    lib().vo_depth_handoff(pp, ph, pw, int(H), int(W), int(bool(invert)), out.ctypes.data_as(_u8p))
    return out


# ------------------------------------------------------------------------------ composite
def pixel_shift(rgb_chw, depth, W, H, params: ShiftParams, state: State | None = None,
                want_shift=False, want_dshaped=False):
    """Restatement of pixel_shift_cuda.  Returns dict(left,right[,shift,dshaped],dbg)."""
    rgb, pr = _f(rgb_chw)
    d, pd = _f(depth)
    in_h, in_w = rgb.shape[1:]
    st = state if state is not None else State()
    L = np.empty((H, W, 3), np.uint8)
    R = np.empty((H, W, 3), np.uint8)
    S = np.empty((1, H, W), np.float32) if want_shift else None
    Dm = np.empty((1, H, W), np.float32) if want_dshaped else None
    dbg = ShiftDebug()
    lib().vo_pixel_shift(pr, pd, in_h, in_w, int(W), int(H), C.byref(params), C.byref(st),
                         L.ctypes.data_as(_u8p), R.ctypes.data_as(_u8p),
                         S.ctypes.data_as(_f32p) if want_shift else None,
                         Dm.ctypes.data_as(_f32p) if want_dshaped else None, C.byref(dbg))
    out = {"left": L, "right": R, "dbg": {k: getattr(dbg, k) for k, _ in ShiftDebug._fields_}}
    if want_shift:
        out["shift"] = S
    if want_dshaped:
        out["dshaped"] = Dm
    return out


def finish_frame(L, R, depth_norm, params: RenderParams, focal_depth, bar_width=0, bar_side=0):
    L, pl = _u(L)
    R, pr = _u(R)
    dn, pd = _f(depth_norm)
    eh, ew = dn.shape[-2:]
    out = np.empty((params.out_h, params.out_w, 3), np.uint8)
    rc = lib().vo_finish_frame(pl, pr, pd, eh, ew, C.byref(params), float(focal_depth), int(bar_width),
                               int(bar_side), out.ctypes.data_as(_u8p))
    if rc:
        raise NotImplementedError("oracle: unsupported fit/format")
    return out


class RenderOracle:
    """Stateful CPU restatement of the render_sbs_3d loop body (one call per frame)."""

    def __init__(self, params: RenderParams, state: State | None = None):
        self.p = params
        self.state = state if state is not None else State()
        ne = params.eye_h * params.eye_w
        self.tdf_prev = np.zeros(ne, np.float32)
        self.norm_prev = np.zeros(ne, np.float32)
        self.last = FrameScalars()

    def new_clip(self):
        """What render_sbs_3d re-creates per call (:1174-1182); the module singletons persist."""
        s = self.state
        s.smooth_valid = 0
        s.tdf_valid = 0
        s.prev_depth_valid = 0
        s.focal_valid = 0

    def render(self, frame_bgr, depth, depth_fmt, want_eyes=False, blank=False):
        """blank=True: the frame is in the skip_blank_frames set (core/render_3d.py:1278-1281)."""
        p = self.p
        fb, pf = _u(frame_bgr)
        if depth_fmt == 0:
            dd = np.ascontiguousarray(depth, dtype=np.float32)
        else:
            dd = np.ascontiguousarray(depth, dtype=np.uint8)
        out = np.empty((p.out_h, p.out_w, 3), np.uint8)
        if blank:
            rc = lib().vo_render_frame_blank(pf, dd.ctypes.data_as(C.c_void_p), int(depth_fmt), C.byref(p),
                                             C.byref(self.state), self.tdf_prev.ctypes.data_as(_f32p),
                                             self.norm_prev.ctypes.data_as(_f32p), out.ctypes.data_as(_u8p),
                                             C.byref(self.last))
            if rc:
                raise NotImplementedError("oracle: unsupported fit/format")
            return (out, None, None) if want_eyes else out
        L = np.empty((p.warp_h, p.warp_w, 3), np.uint8) if want_eyes else None
        R = np.empty((p.warp_h, p.warp_w, 3), np.uint8) if want_eyes else None
        rc = lib().vo_render_frame(pf, dd.ctypes.data_as(C.c_void_p), int(depth_fmt), C.byref(p),
                                   C.byref(self.state), self.tdf_prev.ctypes.data_as(_f32p),
                                   self.norm_prev.ctypes.data_as(_f32p), out.ctypes.data_as(_u8p),
                                   C.byref(self.last),
                                   L.ctypes.data_as(_u8p) if want_eyes else None,
                                   R.ctypes.data_as(_u8p) if want_eyes else None)
        if rc:
            raise NotImplementedError("oracle: unsupported fit/format")
        return (out, L, R) if want_eyes else out


def resize_linear_u8(src, dh, dw):
    """cv2.resize(src, (dw, dh)) (INTER_LINEAR, the default) on uint8 [h,w,3] (unpinned: OpenCV's published fixed-point algorithm)."""
    s = np.ascontiguousarray(src, np.uint8)
    out = np.empty((dh, dw, 3), np.uint8)
    lib().vo_resize_linear_u8(s.ctypes.data_as(_u8p), s.shape[0], s.shape[1], out.ctypes.data_as(_u8p), int(dh), int(dw))
    return out


def resize_cubic_u8(src, dh, dw):
    """cv2.resize(src, (dw, dh), interpolation=cv2.INTER_CUBIC) on uint8 [h,w] or [h,w,3] (unpinned: OpenCV's published algorithm)."""
    s = np.ascontiguousarray(src, np.uint8)
    cn = 1 if s.ndim == 2 else s.shape[2]
    out = np.empty((dh, dw) + (() if s.ndim == 2 else (cn,)), np.uint8)
    lib().vo_resize_cubic_u8(s.ctypes.data_as(_u8p), s.shape[0], s.shape[1], cn, out.ctypes.data_as(_u8p), int(dh), int(dw))
    return out


def esr_pre(bgr):
    """preprocess_esr (core/merged_pipeline.py:219-223) without the batch axis: [h,w,3] BGR uint8 -> [3,h,w] RGB float32."""
    s = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty((3,) + s.shape[:2], np.float32)
    lib().vo_esr_pre(s.ctypes.data_as(_u8p), s.shape[0], s.shape[1], out.ctypes.data_as(_f32p))
    return out


def esr_post(chw):
    """postprocess_esr (core/merged_pipeline.py:225-229) without the batch axis: [3,h,w] RGB float32 -> [h,w,3] BGR uint8."""
    t, pt = _f(chw)
    out = np.empty(t.shape[1:] + (3,), np.uint8)
    lib().vo_esr_post(pt, t.shape[1], t.shape[2], out.ctypes.data_as(_u8p))
    return out


def add_weighted_u8(a, alpha, b, beta, gamma=0.0):
    """cv2.addWeighted(a, alpha, b, beta, gamma) on uint8 (unpinned: float32 fma form, round half to even)."""
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    out = np.empty_like(a)
    lib().vo_add_weighted_u8(a.ctypes.data_as(_u8p), C.c_double(alpha), b.ctypes.data_as(_u8p), C.c_double(beta), C.c_double(gamma),
                             C.c_longlong(a.size), out.ctypes.data_as(_u8p))
    return out


def preview_heatmap(kind, shift, lut):
    """The colour-mapped previews of generate_preview_image (core/preview_utils.py:42-66): kind 0 shift, 1 |shift|, 2 clipped, 3 feather
    mask; ``lut``: uint8 [256,3] BGR."""
    s, ps = _f(np.squeeze(shift))
    lut = np.ascontiguousarray(lut, np.uint8).reshape(256, 3)
    out = np.empty(s.shape + (3,), np.uint8)
    rc = lib().vo_preview_heatmap(int(kind), ps, s.shape[0], s.shape[1], lut.ctypes.data_as(_u8p), out.ctypes.data_as(_u8p))
    assert rc == 0
    return out


def nv12_to_bgr(nv12):
    """uint8 [h*3/2, w] NV12 -> uint8 BGR [h,w,3] (BT.601 limited range, 20-bit fixed point; no reference counterpart)."""
    t = np.ascontiguousarray(nv12, np.uint8)
    h, w = t.shape[0] * 2 // 3, t.shape[1]
    out = np.empty((h, w, 3), np.uint8)
    lib().vo_nv12_to_bgr(t.ctypes.data_as(_u8p), h, w, out.ctypes.data_as(_u8p))
    return out


def bgr_to_nv12(bgr):
    t = np.ascontiguousarray(bgr, np.uint8)
    h, w = t.shape[:2]
    out = np.empty((h * 3 // 2, w), np.uint8)
    lib().vo_bgr_to_nv12(t.ctypes.data_as(_u8p), h, w, out.ctypes.data_as(_u8p))
    return out


def preview_arrows(left, shift):
    """ "Overlay Arrows" of generate_preview_image (core/preview_utils.py:74-82); unpinned (cv2.arrowedLine restated)."""
    L, pl = _u(left)
    s, ps = _f(np.squeeze(shift))
    out = np.empty_like(L)
    lib().vo_preview_arrows(pl, ps, L.shape[0], L.shape[1], out.ctypes.data_as(_u8p))
    return out
