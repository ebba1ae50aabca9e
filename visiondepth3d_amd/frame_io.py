"""Frame I/O boundary: host frames in, muxed frames out, without stalling the GPU.

The reference reads frames with ``cv2.VideoCapture`` into pageable NumPy arrays, uploads them synchronously
(``frame_to_tensor`` ... ``.to(device)``, core/render_3d.py:135-138,1222-1228) and downloads every result synchronously
(``tensor_to_frame`` ... ``.cpu()``, :289-291) before handing it to ``cv2.VideoWriter`` / the ffmpeg ``bgr24`` pipe
(:1143-1163,1422-1427).  Here the same host-side contract (uint8 BGR arrays in, uint8 BGR arrays out) runs through a ring of
PINNED staging buffers and two copy streams, so H2D of batch i+1 and D2H of batch i-1 overlap the kernels of batch i.
Decode / encode themselves stay with the caller (container I/O is out of scope).
"""
from __future__ import annotations

import numpy as np
import torch


class PinnedRing:
    """Ring of ``depth`` slots; each slot = pinned host input / device input / device output / pinned host output."""

    def __init__(self, batch: int, in_shape, out_shape, device, depth: int = 3, extra_in=None):
        self.n, self.B, self.device = depth, batch, torch.device(device)
        mk = lambda shp, dt, pin: (torch.empty((batch,) + tuple(shp), dtype=dt).pin_memory() if pin
                                   else torch.empty((batch,) + tuple(shp), dtype=dt, device=self.device))
        self.h_in = [mk(in_shape, torch.uint8, True) for _ in range(depth)]
        self.d_in = [mk(in_shape, torch.uint8, False) for _ in range(depth)]
        self.d_out = [mk(out_shape, torch.uint8, False) for _ in range(depth)]
        self.h_out = [mk(out_shape, torch.uint8, True) for _ in range(depth)]
        self.h_x = self.d_x = None
        if extra_in is not None:   # second input plane per frame (the depth video frame), (shape, dtype)
            shp, dt = extra_in
            self.h_x = [mk(shp, dt, True) for _ in range(depth)]
            self.d_x = [mk(shp, dt, False) for _ in range(depth)]
        self.s_in, self.s_out = torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)
        self.ev_in = [torch.cuda.Event() for _ in range(depth)]
        self.ev_done = [torch.cuda.Event() for _ in range(depth)]
        self.ev_out = [torch.cuda.Event() for _ in range(depth)]
        self.used = [False] * depth
        self.staged = [False] * depth   # an H2D copy out of h_in[k] / h_x[k] has been enqueued at least once

    def upload(self, k: int, frames_host, extra_host=None, compute_stream=None):
        """Stage (host memcpy into pinned memory) and enqueue the H2D copy of slot k; the compute stream waits for it."""
        if self.staged[k]:
            self.ev_in[k].synchronize()               # HOST: the previous H2D out of the pinned slot has executed before it is restaged
        if self.used[k]:
            self.s_in.wait_event(self.ev_done[k])     # the previous occupant of d_in[k] has been consumed by its kernels
        def stage(src, pinned_slot):   # a decoder that writes into pinned memory itself skips the staging memcpy
            t = src if torch.is_tensor(src) else torch.from_numpy(np.ascontiguousarray(src))
            if t.is_pinned():
                return t
            pinned_slot.copy_(t)
            return pinned_slot
        src_f = stage(frames_host, self.h_in[k])
        src_x = stage(extra_host, self.h_x[k]) if (self.h_x is not None and extra_host is not None) else None
        with torch.cuda.stream(self.s_in):
            self.d_in[k].copy_(src_f, non_blocking=True)
            if src_x is not None:
                self.d_x[k].copy_(src_x, non_blocking=True)
            self.ev_in[k].record(self.s_in)
        self.staged[k] = True
        (compute_stream or torch.cuda.current_stream(self.device)).wait_event(self.ev_in[k])

    def download(self, k: int, compute_stream=None):
        """Call after the kernels of slot k were enqueued: D2H of d_out[k] on the copy-out stream."""
        cs = compute_stream or torch.cuda.current_stream(self.device)
        self.ev_done[k].record(cs)
        self.s_out.wait_event(self.ev_done[k])
        with torch.cuda.stream(self.s_out):
            self.h_out[k].copy_(self.d_out[k], non_blocking=True)
            self.ev_out[k].record(self.s_out)
        self.used[k] = True

    def result(self, k: int) -> torch.Tensor:
        """Block until slot k's output is in host memory; the returned pinned tensor is valid until slot k is reused."""
        self.ev_out[k].synchronize()
        return self.h_out[k]

    def reserve_output(self, k: int, compute_stream=None):
        """Before writing d_out[k] again: its previous content must have left for the host."""
        if self.used[k]:
            (compute_stream or torch.cuda.current_stream(self.device)).wait_event(self.ev_out[k])


def render_clip_pipelined(renderer, frames, depths, params, depth: int = 3):
    """``render_clip`` with the pinned ring: frames / depths are iterables of host arrays (uint8 BGR frame, depth-video frame or
    float32 depth); yields host uint8 BGR muxed frames (copies).  Same read order as the reference: the first frame of the clip
    is consumed and never rendered."""
    it = iter(zip(frames, depths))
    first = next(it, None)
    if first is None:
        return
    f0, d0 = np.asarray(first[0]), np.asarray(first[1])
    ring = PinnedRing(1, f0.shape, (params.out_h, params.out_w, 3), renderer.device, depth,
                      extra_in=(d0.shape, torch.from_numpy(d0).dtype))
    renderer.new_clip()
    pending = []
    for i, (f, d) in enumerate(it):
        k = i % depth
        if len(pending) == depth:
            yield ring.result(pending.pop(0))[0].numpy().copy()
        ring.upload(k, np.asarray(f)[None], np.asarray(d)[None])
        ring.reserve_output(k)
        renderer.render_frame(ring.d_in[k][0], ring.d_x[k][0], params, out=ring.d_out[k][0])
        ring.download(k)
        pending.append(k)
    for k in pending:
        yield ring.result(k)[0].numpy().copy()
