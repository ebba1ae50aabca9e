"""Procedural synthetic clips: platform-independent by construction.

Everything is integer arithmetic (wrapping uint32 hashes, triangle waves) or a handful of
IEEE float64 +,*,/ operations, so the development container and the GPU box generate
bit-identical inputs -- the golden fixtures only need to store expected OUTPUTS.

frame : uint8  [h, w, 3] BGR   low-frequency gradients + 6 moving shapes + +-8 noise
depth : float32 [h, w] in [0,1] ramp .15 -> .85 top->bottom, the same shapes as near blobs
        (.20-.40, hard edges), +-0.01 noise.  Guarantees hi-lo >> 1e-5, >=20 valid samples,
        values on both sides of .05/.95 after normalisation, real depth edges.
"""
from __future__ import annotations

import numpy as np

SEED0 = 0xD1B2

# (kind, cx/1000 of w, cy/1000 of h, rx/1000 of w, ry/1000 of h, dx px/frame, dy px/frame, depth*100, B, G, R)
_SHAPES = (
    (0, 180, 300, 90, 160, 2, 0, 22, 40, 60, 220),
    (1, 520, 520, 120, 200, -2, 0, 30, 200, 180, 40),
    (0, 760, 250, 70, 110, 2, 1, 26, 30, 200, 90),
    (1, 330, 720, 100, 120, 2, -1, 36, 230, 230, 230),
    (0, 880, 700, 60, 180, -2, 0, 20, 20, 20, 20),
    (1, 610, 170, 50, 80, 2, 0, 40, 120, 40, 160),
)


def _hash(x, y, t, c):
    """xorshift-multiply mix on uint32 (wrapping), returns uint32 array."""
    with np.errstate(over="ignore"):
        h = (x.astype(np.uint32) * np.uint32(73856093)) ^ (y.astype(np.uint32) * np.uint32(19349663))
        h ^= np.uint32((int(t) * 83492791 + int(c) * 2654435761 + SEED0) & 0xFFFFFFFF)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0x5BD1E995)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x27D4EB2D)
        h ^= h >> np.uint32(16)
    return h


def _tri(t, period, amp):
    """integer triangle wave in [-amp, amp]."""
    u = np.mod(t, period)
    v = np.abs(u - period // 2)
    return (v * (2 * amp)) // (period // 2) - amp


def synth_frame(idx: int, h: int, w: int):
    """Return (bgr uint8 [h,w,3], depth float32 [h,w]) for frame ``idx``."""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    s = max(w // 64, 1)  # pattern scale so 4K and 144p look alike
    chans = []
    for c, (a, b, per, base) in enumerate(((3, 1, 97, 110), (1, 2, 131, 120), (2, 3, 173, 100))):
        t = (a * x + b * y) // s + (c + 1) * idx
        chans.append(base + _tri(t, per, 55))
    mask_depth = np.zeros((h, w), np.int64)  # depth*100 of the nearest shape, 0 = none
    for (kind, cx, cy, rx, ry, dx, dy, d100, cb, cg, cr) in _SHAPES:
        px = (cx * w // 1000 + dx * idx * s) % w
        py = (cy * h // 1000 + dy * idx * s) % h
        ax, ay = max(rx * w // 1000, 2), max(ry * h // 1000, 2)
        ddx = np.abs(x - px)
        ddx = np.minimum(ddx, w - ddx)  # wrap horizontally so shapes re-enter
        ddy = np.abs(y - py)
        if kind == 0:
            m = (ddx <= ax) & (ddy <= ay)
        else:
            m = (ddx * ddx) * (ay * ay) + (ddy * ddy) * (ax * ax) <= (ax * ax) * (ay * ay)
        for ch, col in zip(chans, (cb, cg, cr)):
            ch[m] = col
        mask_depth[m] = d100
    bgr = np.empty((h, w, 3), np.uint8)
    for c in range(3):
        noise = (_hash(x, y, idx, c) % np.uint32(17)).astype(np.int64) - 8
        bgr[..., c] = np.clip(chans[c] + noise, 0, 255).astype(np.uint8)
    ramp = 0.15 + 0.70 * (y.astype(np.float64) / float(max(h - 1, 1)))
    d = np.where(mask_depth > 0, mask_depth.astype(np.float64) / 100.0, ramp)
    dn = ((_hash(x, y, idx, 7) % np.uint32(2001)).astype(np.float64) - 1000.0) / 100000.0
    depth = np.clip(d + dn, 0.0, 1.0).astype(np.float32)
    return bgr, depth


def depth_to_u8_bgr(depth: np.ndarray) -> np.ndarray:
    """What a depth *video* frame holds: gray = trunc(d*255) replicated to BGR (render_depth.py:608-611,1932)."""
    g = (depth.astype(np.float32) * np.float32(255)).astype(np.uint8)
    return np.repeat(g[..., None], 3, axis=2)


def synth_clip(n: int, h: int, w: int, start: int = 0):
    frames, depths = [], []
    for i in range(start, start + n):
        f, d = synth_frame(i, h, w)
        frames.append(f)
        depths.append(d)
    return frames, depths


def letterbox_clip(n: int, sh: int, sw: int, top: int, bottom: int):
    """synth clip with dark bars painted over it (auto_crop_black_bars fixtures): bar pixels = (3*x + y + 5*t) % 10, so
    bar-row means stay <= 9 < the detector's threshold 10; frame 2 additionally has a dark first content row (the detected
    top moves by one on that frame).  Returns (frames_bgr_u8, depth_bgr_u8) lists."""
    frames, depths = synth_clip(n, sh, sw)
    y, x = np.mgrid[0:sh, 0:sw]
    out_f, out_d = [], []
    for t, (f, d) in enumerate(zip(frames, depths)):
        f = f.copy()
        bar = ((3 * x + y + 5 * t) % 10).astype(np.uint8)
        m = (y < top) | (y >= sh - bottom)
        f[m] = bar[m][:, None]
        if t == 2:
            f[top] = 7
        db = depth_to_u8_bgr(d).copy()
        db[m] = 0
        out_f.append(f)
        out_d.append(db)
    return out_f, out_d
