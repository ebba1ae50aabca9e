"""Frame sharding of ONE clip across GPUs, bit-identical to the sequential render.

The reference's per-frame path carries sequential temporal state: a plane EMA (TemporalDepthFilter, core/render_3d.py:220-229),
a percentile EMA (:233-262) and non-linear scalar trackers (:463-511,895-922).  Frames therefore cannot simply be dealt out --
but everything expensive (depth inference, the select chain, warp, DOF, sharpen, mux) can, once the STATE is exchanged:

  step = world x B consecutive frames, cut into world CONTIGUOUS chunks; rank g owns frames [g*B, (g+1)*B) of the step

    0. (caller) depth inference for the B own frames                                              parallel, MFMA-bound
    1. chunk hand-off, point to point: rank g receives the filtered depth plane of the frame before its chunk from rank g-1
       (ONE eye-size float32 plane per chunk boundary: 2.1 MB @1080p Half-SBS, 8.3 MB @4K), runs P1 over its own frames in
       order (ingest + plane EMA + exact q.02 / q.98) and sends its last plane on to rank g+1 (rank world-1 -> rank 0 of the
       next step).  This is the only serial part: the plane EMA is a float32 recurrence, not truncatable bit-exactly.
    2. all-gather of (q_lo, q_hi) per frame (8 B x frames); R1: DepthPercentileEMA replayed on every rank
    3. P3 per own frame: normalise, centre-crop sums, MAD, subject depths, the warp-res select chain, shaped depth, s1
    4. all-gather of 4 x int64 per frame; R2: dynamic parallax scale, ShiftSmoother, FocalDepthTracker, FloatingWindowTracker,
       ConvergenceEMA, FloatingBarEaser replayed in frame order on every rank (incl. the skip_blank_frames flags); own slots patched
    5. pixel pass (shift plane, fused warp, fused finish) per own frame

Data-path traffic per step and rank: 40 B per frame of records (two small all-gathers) + one plane per chunk boundary.  Every
reduction is integer / fixed-point and every replay is sequential scalar code, so all ranks hold bit-identical tracker state.

``ChunkSharder`` is backend-agnostic: ``HipChunkBackend`` drives libvd3d_hip.so; tests/test_sharded_gloo.py runs the same
orchestration over gloo with the CPU oracle as backend (numerics, not a recording fake).
"""
from __future__ import annotations

import time

import torch
import torch.distributed as dist


class HipChunkBackend:
    """The protocol's stages on one ``visiondepth3d_amd.render_3d.Renderer`` (include/vd3d.h ``vd3d_shard2_*``)."""

    def __init__(self, renderer, params):
        self.r, self.p = renderer, params
        self.device = renderer.device
        self.auto_crop = bool(params.auto_crop_black_bars)

    def begin(self, n_slots: int):
        self.r.shard_begin(self.p, n_slots)

    def new_clip(self):
        self.r.new_clip()

    def plane_shape(self):
        return (self.p.eye_h, self.p.eye_w)

    def plane_export(self, out=None):
        return self.r.tdf_plane_export(self.p, out)

    def plane_import(self, plane, valid=True):
        self.r.tdf_plane_import(plane, self.p, valid)

    def p0(self, frame, crop_out):
        self.r.shard2_p0(frame, self.p, crop_out)

    def set_crops(self, crops_all):
        self.r.shard2_set_crops(crops_all)

    def p1(self, frame, depth, step_idx, slot, q_out):
        self.r.shard2_p1(frame, depth, self.p, step_idx, slot, q_out)

    MAX_BATCH = 16   # frames per batched launch (VD_MAX_BATCH)

    def p1_batch(self, frames, depths, step_idx0, slot0, q_out):
        """P1 of consecutive own frames: two launches per <= 16 frames instead of two per frame (the plane EMA walks the frames inside
        the ingest kernel; the quantile passes of the frames run side by side)."""
        for j0 in range(0, len(frames), self.MAX_BATCH):
            j1 = min(len(frames), j0 + self.MAX_BATCH)
            self.r.shard2_p1_batch(frames[j0:j1], depths[j0:j1], self.p, step_idx0 + j0, slot0 + j0, q_out[j0:j1])

    def p3_batch(self, slot0, step_idx0, n, m_out):
        for j0 in range(0, n, self.MAX_BATCH):
            j1 = min(n, j0 + self.MAX_BATCH)
            self.r.shard2_p3_batch(slot0 + j0, step_idx0 + j0, j1 - j0, self.p, m_out[j0:j1])

    def r1(self, q_all):
        self.r.shard2_r1(q_all)

    def p3(self, slot, step_idx, m_out):
        self.r.shard2_p3(slot, step_idx, self.p, m_out)

    def r2(self, m_all, own_slots, blank):
        self.r.shard2_r2(m_all, own_slots, self.p, blank)

    def pixels(self, slot, out=None, blank_frame=None):
        return self.r.shard_pixels(slot, self.p, out=out, blank_frame=blank_frame)


class ChunkSharder:
    """Contiguous-chunk frame sharding (module docstring).  ``backend``: HipChunkBackend or anything with the same methods.

    One instance = one set of ``B`` slots; two instances with ``slot_base`` 0 and B on one renderer keep two steps in flight
    (the pixel pass of step i after the measurements of step i+1 were enqueued: ``vd3d_set_pixel_overlap``, bench.py)."""

    def __init__(self, backend, rank: int, world: int, frames_per_rank: int, group=None, slot_base: int = 0, twin_of=None):
        """``twin_of``: the sharder of the other slot set on the same backend (they share the chunk hand-off state)."""
        self.b, self.rank, self.world, self.B, self.group = backend, int(rank), int(world), int(frames_per_rank), group
        self._shared = twin_of._shared if twin_of is not None else {"pending": None}   # rank 0: plane received from rank world-1
        self._wait = []   # P1-chain waits of this slot set: (event, event) pairs on the GPU, seconds on the CPU (p1_wait_ms)
        self.slot_base = int(slot_base)
        if world * frames_per_rank > 512:
            raise ValueError("a sharded step holds at most 512 frames")
        backend.begin(self.slot_base + frames_per_rank)
        dev = backend.device
        self.q_local = torch.zeros((frames_per_rank, 2), dtype=torch.float32, device=dev)
        self.m_local = torch.zeros((frames_per_rank, 4), dtype=torch.int64, device=dev)
        self.crops = torch.zeros((world * frames_per_rank, 4), dtype=torch.int32, device=dev)
        self.auto_crop = bool(getattr(backend, "auto_crop", False))
        # frame t of the step belongs to rank t // B; this rank's frames sit in slots slot_base .. slot_base + B - 1
        self.own_slots = [(self.slot_base + t % self.B if t // self.B == self.rank else -1) for t in range(world * frames_per_rank)]

    # ---- collectives / point to point -------------------------------------------------------------------------------------
    def gather(self, local: torch.Tensor) -> torch.Tensor:
        """all-gather along dim 0: [B, ...] per rank -> [world*B, ...]; rank-major order IS frame order for contiguous chunks."""
        if self.world == 1:
            return local
        out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def bytes_per_step(self) -> dict:
        """Communication of ONE step per rank, for the benchmark record (so a scaling run can be checked against the protocol):
        one eye-size float32 plane sent and one received per chunk boundary (none at world 1), two all-gathers of the per-frame
        records (q: 2 x float32, m: 4 x int64 per frame of the step)."""
        n = self.world * self.B
        ph, pw = self.b.plane_shape()
        return {"ranks": self.world, "frames_per_step": n, "p2p_plane_bytes_sent": 0 if self.world == 1 else int(ph) * int(pw) * 4,
                "p2p_plane_bytes_received": 0 if self.world == 1 else int(ph) * int(pw) * 4,
                "allgather_bytes_received": 0 if self.world == 1 else n * (2 * 4 + 4 * 8),
                "collectives_per_step": 0 if self.world == 1 else 2, "backend": (("nccl (RCCL over xGMI)" if dist.get_backend(self.group) == "nccl" else dist.get_backend(self.group)) if self.world > 1 else None)}

    def _send(self, t, dst):
        dist.send(t.contiguous(), dst, group=self.group)

    def _recv(self, t, src):
        dist.recv(t, src, group=self.group)

    def _recv_plane(self, t, src):
        """The chunk hand-off receive, timed: how long this rank's stream sits behind the previous chunk's P1 (the serial part of the
        protocol) plus the plane transfer itself.  GPU: an event pair on the current stream around the receive; CPU: wall clock."""
        if t.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._recv(t, src)
            e1.record()
            self._wait.append((e0, e1))
        else:
            t0 = time.perf_counter()
            self._recv(t, src)
            self._wait.append(time.perf_counter() - t0)
        del self._wait[:-64]

    def p1_wait_ms(self):
        """Mean milliseconds per step this rank waited for (and received) the plane of the previous chunk; None if it never received."""
        if not self._wait:
            return None
        if isinstance(self._wait[0], tuple):
            self._wait[-1][1].synchronize()
            v = [a.elapsed_time(b) for a, b in self._wait]
        else:
            v = [1e3 * w for w in self._wait]
        return round(sum(v) / len(v), 4)

    def _nv(self, n_valid):
        n = self.world * self.B if n_valid is None else int(n_valid)
        if not (1 <= n <= self.world * self.B):
            raise ValueError("n_valid out of range")
        return n

    def own_range(self, n_valid=None):
        """Local indices j (0..B-1) of this rank's frames that exist in a step with ``n_valid`` frames."""
        n = self._nv(n_valid)
        return range(max(0, min(self.B, n - self.rank * self.B)))

    # ---- the stages ---------------------------------------------------------------------------------------------------------
    def p1(self, frames_local, depths_local, n_valid=None, first_step=False, more_steps=True):
        """Chunk hand-off + P1 over the own frames.  ``first_step``: the clip starts here (rank 0 keeps its own, fresh state);
        ``more_steps``: another step follows (rank world-1 then hands its plane to rank 0)."""
        G, g = self.world, self.rank
        own = self.own_range(n_valid)
        if G > 1:
            # a plane that is handed on is always a valid one: its sender has rendered (or received) at least one frame of the clip
            if g > 0:
                buf = torch.empty(self.b.plane_shape(), dtype=torch.float32, device=self.b.device)
                self._recv_plane(buf, g - 1)
                self.b.plane_import(buf, valid=True)
            elif not first_step and self._shared["pending"] is not None:
                self.b.plane_import(self._shared["pending"], valid=True)
                self._shared["pending"] = None
        self.p1_local(frames_local, depths_local, n_valid)
        if G > 1:
            last = g == G - 1
            if not last or more_steps:
                self._send(self.b.plane_export(), (g + 1) % G)
            if g == 0 and more_steps:   # posted BEFORE the all-gathers of this step (stream-ordered backends would deadlock otherwise)
                buf = torch.empty(self.b.plane_shape(), dtype=torch.float32, device=self.b.device)
                self._recv(buf, G - 1)
                self._shared["pending"] = buf

    def p1_local(self, frames_local, depths_local, n_valid=None):
        """The compute half of ``p1`` (no communication): P1 over the own frames, the plane state being already in place."""
        g = self.rank
        own = self.own_range(n_valid)
        if self.auto_crop:   # black-bar detection per own frame; the rectangle table only needs this rank's rows
            self.crops.zero_()
            for j in own:
                self.b.p0(frames_local[j], self.crops[g * self.B + j])
            self.b.set_crops(self.crops[:self._nv(n_valid)])
        if hasattr(self.b, "p1_batch") and len(own):   # own frames are consecutive: batched launches
            self.b.p1_batch([frames_local[j] for j in own], [depths_local[j] for j in own], g * self.B + own[0], self.slot_base + own[0],
                            self.q_local)
        else:
            for j in own:
                self.b.p1(frames_local[j], depths_local[j], g * self.B + j, self.slot_base + j, self.q_local[j])

    def r1(self, q_all, n_valid=None):
        self.b.r1(q_all[:self._nv(n_valid)])

    def p3(self, n_valid=None):
        own = self.own_range(n_valid)
        if hasattr(self.b, "p3_batch") and len(own):
            self.b.p3_batch(self.slot_base + own[0], self.rank * self.B + own[0], len(own), self.m_local)
            return
        for j in own:
            self.b.p3(self.slot_base + j, self.rank * self.B + j, self.m_local[j])

    def r2(self, m_all, n_valid=None, blank=None):
        n = self._nv(n_valid)
        self.b.r2(m_all[:n], self.own_slots[:n], None if blank is None else list(blank)[:n])

    def pixels(self, outs=None, n_valid=None, frames_local=None, blank=None):
        """Pixel pass of the own frames; ``blank``: per-step flags (frame order) -- blank own frames need ``frames_local``."""
        res = []
        for j in self.own_range(n_valid):
            t = self.rank * self.B + j
            bf = frames_local[j] if (blank is not None and blank[t]) else None
            res.append(self.b.pixels(self.slot_base + j, out=None if outs is None else outs[j], blank_frame=bf))
        return res

    def render_step(self, frames_local, depths_local, outs=None, n_valid=None, blank=None, first_step=False, more_steps=True):
        """One step.  ``blank``: None or world*B flags in frame order (skip_blank_frames set, known to every rank).  Returns the
        muxed frames of this rank's valid own frames."""
        self.p1(frames_local, depths_local, n_valid, first_step, more_steps)
        self.r1(self.gather(self.q_local), n_valid)
        self.p3(n_valid)
        self.r2(self.gather(self.m_local), n_valid, blank)
        return self.pixels(outs, n_valid, frames_local, blank)

    def finish_clip(self):
        """After the last step: every rank ends with the plane state a sequential render would leave behind (the last chunk's
        owner broadcasts its filtered plane), so a later sequential frame or clip continues identically on any rank."""
        if self.world == 1:
            return
        src = self.world - 1
        buf = self.b.plane_export() if self.rank == src else torch.empty(self.b.plane_shape(), dtype=torch.float32, device=self.b.device)
        dist.broadcast(buf, src, group=self.group)
        if self.rank != src:
            self.b.plane_import(buf, valid=True)

    def render_clip(self, n_frames: int, get_frame, get_depth, blank_frames=None, overlap_pixels: bool = False):
        """Render frames 0..n_frames-1 of one clip (callables return the uint8 BGR frame / depth plane of frame t).  Yields
        (t, muxed_frame) for the frames this rank owns, in increasing t; the last step may be partial.  Every rank calls this
        with the same n_frames / blank_frames (the collectives are matched).  ``blank_frames``: frame indices in the
        skip_blank_frames set.

        ``overlap_pixels`` (HIP backend): steps alternate between two slot sets and the pixel kernels of step i run on the
        renderer's second stream while the measurement chain of step i+1 runs on the first; the frames of step i are yielded once
        step i+1 has been enqueued and their pixel passes have completed."""
        G, B = self.world, self.B
        per_step = G * B
        blank_set = set(blank_frames or ())
        sets = [self]
        if overlap_pixels:
            if getattr(self, "_twin", None) is None:
                self._twin = ChunkSharder(self.b, self.rank, G, B, self.group, slot_base=self.slot_base + B, twin_of=self)
            sets = [self, self._twin]
            self.b.r.set_pixel_overlap(True)
        self.b.new_clip()   # collective-safe: every rank starts the clip from the same fresh per-clip state
        self._shared["pending"] = None

        def drain(step):
            own, outs, sh = step
            for j, (t, o) in enumerate(zip(own, outs)):
                self.b.r.wait_pixels(sh.slot_base + j)
                yield t, o

        pending = None
        n_steps = (n_frames + per_step - 1) // per_step
        try:
            for k in range(n_steps):
                base = k * per_step
                nv = min(per_step, n_frames - base)
                sh = sets[k % len(sets)]
                own_j = list(sh.own_range(nv))
                fl = [get_frame(base + self.rank * B + j) for j in own_j]
                dl = [get_depth(base + self.rank * B + j) for j in own_j]
                blank = [(base + t) in blank_set for t in range(nv)] if blank_set else None
                outs = sh.render_step(fl, dl, n_valid=nv, blank=blank, first_step=(k == 0), more_steps=(k + 1 < n_steps))
                own = [base + self.rank * B + j for j in own_j]
                if overlap_pixels:
                    if pending is not None:
                        yield from drain(pending)
                    pending = (own, outs, sh)
                else:
                    for t, o in zip(own, outs):
                        yield t, o
            if pending is not None:
                yield from drain(pending)
            self.finish_clip()
        finally:
            if overlap_pixels:
                self.b.r.set_pixel_overlap(False)   # joins whatever is still in flight
