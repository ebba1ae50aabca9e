"""Stand-in modules so the reference's ``core/render_3d.py`` can be imported in the
development container (which has no cv2 / torchvision / tkinter / onnxruntime).

TEST INFRASTRUCTURE, development container only.  Nothing here is shipped in
the product path and nothing here travels as "the reference": these are
restatements of the *third-party* semantics the reference leans on
(SURVEY.md §8(c) "parity unpinned" list):

* ``cv2.cvtColor``  BGR<->RGB channel reverse; BGR2GRAY / RGB2GRAY in OpenCV's
  14-bit fixed point  (1868*B + 9617*G + 4899*R + 8192) >> 14.
* ``cv2.filter2D``  u8 source, float32 kernel: float32 accumulation over the
  non-zero taps in row-major order, BORDER_REFLECT_101, saturate_cast<uchar>
  (= round half to even, clamp 0..255).
* ``cv2.resize``    INTER_AREA: identity copy when sizes match; integer
  down-scale = integer box sum * (1/area) in float32 then saturate_cast;
  fractional down-scale = OpenCV's resizeArea_ decimation table in float32.
  INTER_LINEAR (only used by the VR format) in OpenCV's 11-bit fixed point.
* ``torchvision.transforms.functional.gaussian_blur``: 1-D pdf on
  linspace(-(k-1)/2,(k-1)/2,k), exp(-0.5 (x/sigma)^2), normalised, outer
  product, reflect pad, depthwise conv2d -- torchvision's public algorithm.

The goldens produced through these stubs therefore pin the reference's own
Python/torch arithmetic exactly and the third-party arithmetic only up to
these restatements.
"""
from __future__ import annotations

import importlib.machinery
import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- cv2
COLOR_BGR2RGB = 4
COLOR_RGB2BGR = 4
COLOR_BGR2GRAY = 6
COLOR_RGB2GRAY = 7
COLOR_GRAY2BGR = 8
INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_CUBIC = 2
INTER_AREA = 3
CAP_PROP_POS_FRAMES = 1
CAP_PROP_FRAME_WIDTH = 3
CAP_PROP_FRAME_HEIGHT = 4
CAP_PROP_FPS = 5
CAP_PROP_FRAME_COUNT = 7


def cvtColor(img, code):
    img = np.asarray(img)
    if code == COLOR_BGR2RGB:  # same value as RGB2BGR: a channel reverse
        return np.ascontiguousarray(img[..., ::-1])
    if code in (COLOR_BGR2GRAY, COLOR_RGB2GRAY):
        a = img.astype(np.int64)
        if code == COLOR_BGR2GRAY:
            b, g, r = a[..., 0], a[..., 1], a[..., 2]
        else:
            r, g, b = a[..., 0], a[..., 1], a[..., 2]
        if img.dtype == np.uint8:
            return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)
        return (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).astype(img.dtype)
    if code == COLOR_GRAY2BGR:
        return np.repeat(img[..., None], 3, axis=2)
    raise NotImplementedError(f"cvtColor code {code}")


def _rne_u8(x):
    """saturate_cast<uchar>(float): round half to even, clamp."""
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def filter2D(src, ddepth, kernel):
    src = np.asarray(src)
    assert src.dtype == np.uint8 and ddepth == -1
    k = np.asarray(kernel, dtype=np.float32)
    kh, kw = k.shape
    ay, ax = kh // 2, kw // 2
    pad = np.pad(src, ((ay, kh - 1 - ay), (ax, kw - 1 - ax), (0, 0)), mode="reflect").astype(np.float32)
    H, W = src.shape[:2]
    acc = np.zeros(src.shape, dtype=np.float32)
    for i in range(kh):
        for j in range(kw):
            if k[i, j] == 0:
                continue
            acc = (acc + k[i, j] * pad[i:i + H, j:j + W]).astype(np.float32)
    return _rne_u8(acc)


def _area_tab(ssize, dsize, scale):
    """OpenCV computeResizeAreaTab: list of (di, si, alpha float32)."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = int(math.ceil(fsx1))
        sx2 = int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _resize_area(src, dw, dh):
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src.copy()
    sx, sy = 1.0 / (dw / sw), 1.0 / (dh / sh)   # hal::resize: scale = 1./inv_scale, inv_scale = (double)dsize/ssize
    isx, isy = int(round(sx)), int(round(sy))
    if sx >= 1 and sy >= 1 and abs(sx - isx) < 2.2e-16 * 4 and abs(sy - isy) < 2.2e-16 * 4:
        # integer-ratio fast path: int sums * float(1/area) -> saturate_cast; the 2 x 2 ratio of 1- / 3- / 4-channel 8-bit images is
        # ResizeAreaFastVec's fast_mode: (s00 + s01 + s10 + s11 + 2) >> 2, i.e. ties round UP where cvRound would round to even
        # (modules/imgproc/src/resize.cpp; the oracle and k_sharp_mux / ff_epilogue have had that rule since round 1, this stand-in had not)
        a = src.astype(np.int64).reshape(dh, isy, dw, isx, -1).sum(axis=(1, 3))
        if isx == 2 and isy == 2 and a.shape[-1] in (1, 3, 4):
            return ((a + 2) >> 2).astype(np.uint8).reshape(dh, dw, *src.shape[2:])
        scale = np.float32(1.0 / (isx * isy))
        return _rne_u8(a.astype(np.float32) * scale)
    if sx < 1 or sy < 1:
        # any up-scaling dimension: hal::resize skips the area paths and runs the INTER_LINEAR machinery with the
        # "area_mode" coefficients (sx = floor(dx*scale), fx = (dx+1) - (sx+1)*inv_scale, fx -= floor(fx))
        return _resize_linear_u8(src, dw, dh, area_mode=True)
    xt, yt = _area_tab(sw, dw, sx), _area_tab(sh, dh, sy)
    s = src.astype(np.float32)
    # horizontal pass into per-source-row buffers, then vertical accumulation (float32)
    hbuf = np.zeros((sh, dw, s.shape[2]), dtype=np.float32)
    for (dx, sxi, a) in xt:
        hbuf[:, dx] = (hbuf[:, dx] + s[:, sxi] * a).astype(np.float32)
    out = np.zeros((dh, dw, s.shape[2]), dtype=np.float32)
    for (dy, syi, b) in yt:
        out[dy] = (out[dy] + hbuf[syi] * b).astype(np.float32)
    return _rne_u8(out)


def _resize_linear_u8(src, dw, dh, area_mode=False):
    """OpenCV INTER_LINEAR for 8-bit: 11-bit fixed-point coefficients, two-stage rounding.  area_mode = the coefficient rule
    hal::resize uses when INTER_AREA is asked to up-scale."""
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src.copy()
    SC = 2048

    def tab(ssz, dsz):
        inv = dsz / ssz
        sc = 1.0 / inv if area_mode else ssz / dsz
        idx = np.zeros(dsz, np.int64)
        a = np.zeros((dsz, 2), np.int64)
        for d in range(dsz):
            if area_mode:
                i = int(math.floor(d * sc))
                f = np.float32((d + 1) - (i + 1) * inv)
                f = np.float32(0) if f <= 0 else np.float32(f - np.float32(math.floor(f)))
            else:
                f = np.float32((d + 0.5) * sc - 0.5)
                i = int(math.floor(f))
                f = np.float32(f - i)
            if i < 0:
                i, f = 0, np.float32(0)
            if i >= ssz - 1:
                i, f = ssz - 1, np.float32(0)
            idx[d] = i
            a[d, 0] = int(np.rint(np.float32((1.0 - f) * SC)))
            a[d, 1] = int(np.rint(np.float32(f * SC)))
        return idx, a

    xi, xa = tab(sw, dw)
    yi, ya = tab(sh, dh)
    s = src.astype(np.int64)
    x1 = np.minimum(xi + 1, sw - 1)
    rows = s[:, xi] * xa[None, :, 0, None] + s[:, x1] * xa[None, :, 1, None]
    y1 = np.minimum(yi + 1, sh - 1)
    r0, r1 = rows[yi], rows[y1]
    out = (((ya[:, 0, None, None] * (r0 >> 4)) >> 16) + ((ya[:, 1, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize(src, dsize, interpolation=INTER_LINEAR):
    src = np.asarray(src)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[..., None]
    dw, dh = int(dsize[0]), int(dsize[1])
    if interpolation == INTER_AREA:
        out = _resize_area(src, dw, dh)
    elif interpolation == INTER_LINEAR:
        out = _resize_linear_u8(src, dw, dh)
    else:
        raise NotImplementedError(f"resize interpolation {interpolation}")
    return out[..., 0] if squeeze else out


def split(img):
    return [np.ascontiguousarray(img[..., i]) for i in range(img.shape[2])]


def absdiff(a, b):
    return np.abs(np.asarray(a).astype(np.int16) - np.asarray(b).astype(np.int16)).astype(np.uint8)


def merge(chs):
    return np.stack(chs, axis=2)


COLORMAP_BONE, COLORMAP_JET = 1, 2


def test_colormap(cmap):
    """A synthetic 256 x 3 table per colour-map id: cv2's own tables are not available here, and what the fixtures pin is the INDEX
    arithmetic of generate_preview_image -- applyColorMap itself is a table look-up."""
    i = np.arange(256, dtype=np.int64)
    return np.stack([(i * 3 + 17 * cmap) & 255, 255 - i, (i * i + cmap) & 255], axis=1).astype(np.uint8)


def applyColorMap(src, cmap):
    return test_colormap(cmap)[np.asarray(src, np.uint8)]


def bitwise_and(a, b):
    return np.bitwise_and(a, b)


def VideoWriter_fourcc(*a):
    return "".join(a)


class _Clip:
    """Registry of in-memory clips addressed by fake path."""
    clips: dict = {}
    written: dict = {}


class VideoCapture:
    def __init__(self, path):
        self.frames = _Clip.clips.get(path)
        self.pos = 0
        self.log = []
        self.fps = 24.0

    def isOpened(self):
        return self.frames is not None

    def get(self, prop):
        if prop == CAP_PROP_FRAME_COUNT:
            return float(len(self.frames))
        if prop == CAP_PROP_FPS:
            return self.fps
        if prop == CAP_PROP_POS_FRAMES:
            return float(self.pos)
        if prop == CAP_PROP_FRAME_WIDTH:
            return float(self.frames[0].shape[1])
        if prop == CAP_PROP_FRAME_HEIGHT:
            return float(self.frames[0].shape[0])
        return 0.0

    def set(self, prop, v):
        if prop == CAP_PROP_POS_FRAMES:
            self.pos = int(v)
        return True

    def read(self):
        if self.pos >= len(self.frames):
            return False, None
        f = self.frames[self.pos].copy()
        self.log.append(self.pos)
        self.pos += 1
        return True, f

    def release(self):
        pass


class VideoWriter:
    def __init__(self, path, fourcc, fps, size):
        self.path = path
        self.size = size
        _Clip.written[path] = []

    def isOpened(self):
        return True

    def write(self, frame):
        _Clip.written[self.path].append(np.array(frame, copy=True))

    def release(self):
        pass


# ------------------------------------------------------------------- torchvision
def _gaussian_kernel1d(kernel_size: int, sigma: float) -> torch.Tensor:
    ksize_half = (kernel_size - 1) * 0.5
    x = torch.linspace(-ksize_half, ksize_half, steps=kernel_size)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(img: torch.Tensor, kernel_size, sigma=None):
    if isinstance(kernel_size, int):
        kernel_size = [kernel_size, kernel_size]
    if isinstance(sigma, (int, float)):
        sigma = [float(sigma), float(sigma)]
    kx = _gaussian_kernel1d(kernel_size[0], sigma[0]).to(img.dtype)
    ky = _gaussian_kernel1d(kernel_size[1], sigma[1]).to(img.dtype)
    k2 = torch.mm(ky[:, None], kx[None, :])
    C = img.shape[-3]
    k2 = k2.expand(C, 1, k2.shape[0], k2.shape[1])
    x = img.unsqueeze(0) if img.dim() == 3 else img
    pad = [kernel_size[0] // 2, kernel_size[0] // 2, kernel_size[1] // 2, kernel_size[1] // 2]
    x = F.pad(x, pad, mode="reflect")
    x = F.conv2d(x, k2, groups=C)
    return x.squeeze(0) if img.dim() == 3 else x


# ------------------------------------------------------------------------ install
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Register the stub modules (idempotent)."""
    if "cv2" in sys.modules and getattr(sys.modules["cv2"], "_vd3d_stub", False):
        return
    g = globals()
    cv2_attrs = {k: g[k] for k in (
        "COLOR_BGR2RGB", "COLOR_RGB2BGR", "COLOR_BGR2GRAY", "COLOR_RGB2GRAY", "COLOR_GRAY2BGR",
        "INTER_NEAREST", "INTER_LINEAR", "INTER_CUBIC", "INTER_AREA",
        "CAP_PROP_POS_FRAMES", "CAP_PROP_FRAME_WIDTH", "CAP_PROP_FRAME_HEIGHT", "CAP_PROP_FPS",
        "CAP_PROP_FRAME_COUNT", "cvtColor", "filter2D", "resize", "split", "merge", "absdiff", "bitwise_and", "COLORMAP_BONE", "COLORMAP_JET", "applyColorMap",
        "VideoWriter_fourcc", "VideoCapture", "VideoWriter")}
    _mod("cv2", _vd3d_stub=True, **cv2_attrs)

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    tk = _mod("tkinter", Tk=_Dummy, TkVersion=8.6, PhotoImage=_Dummy, BitmapImage=_Dummy,
              Label=_Dummy, Toplevel=_Dummy, StringVar=_Dummy)
    tk.filedialog = _mod("tkinter.filedialog")
    tk.messagebox = _mod("tkinter.messagebox")
    _mod("onnxruntime", get_device=lambda: "CPU")
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms")
    tv.transforms.functional = _mod("torchvision.transforms.functional", gaussian_blur=gaussian_blur)
