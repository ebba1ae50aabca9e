"""Frame sharding of ONE clip across GPUs, bit-identical to the sequential render.

The reference's per-frame path carries sequential temporal state (a plane EMA, a percentile EMA and several
non-linear scalar trackers, core/render_3d.py:220-286,463-511,895-922), so frames cannot simply be dealt out.
What CAN be dealt out is the expensive part -- depth inference and the pixel work (warp, DOF, sharpen, mux):

  round r, world G:  rank g owns frame t = r*G + g
    1. every rank runs depth inference for ITS frame                                   (parallel, MFMA-bound)
    2. ONE collective: all-gather of the uint8 depth planes (h*w bytes per frame; 2 MB @1080p) over RCCL/xGMI
       -- skipped entirely when the depth video already exists on every rank
    3. for t in the round, in order: the owner calls render_frame (state advance + pixels),
       every other rank calls advance_state (state advance only: eye-res reductions + scalar stage, no pixels)

Every rank therefore walks the identical state trajectory (the HIP reductions are integer / fixed-point, hence
deterministic across GPUs) and the muxed frames equal the single-GPU render bit for bit.  The only data-path
collective is the depth all-gather; payload per frame is the plane the reference would have written to its
depth video anyway.

The runner is backend-agnostic (``render_frame`` / ``advance_state`` / ``new_clip``) so the orchestration is
covered by world_size-2 gloo tests on CPU with the oracle as backend (tests/test_sharded_gloo.py).
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch
import torch.distributed as dist


class HipBackend:
    """Adapter over visiondepth3d_amd.render_3d.Renderer."""

    def __init__(self, renderer, params):
        self.r, self.p = renderer, params
        self.device = renderer.device

    def new_clip(self):
        self.r.new_clip()

    def render_frame(self, frame, depth):
        return self.r.render_frame(frame, depth, self.p)

    def advance_state(self, depth):
        self.r.advance_state(depth, self.p)


class FrameShardedRenderer:
    def __init__(self, backend, rank: int | None = None, world: int | None = None, group=None):
        self.b = backend
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)

    def new_clip(self):
        self.b.new_clip()

    def owner(self, t: int) -> int:
        return t % self.world

    def render_round(self, n_valid: int, frame=None, depth_local: Optional[torch.Tensor] = None,
                     depth_all: Optional[torch.Tensor] = None):
        """One round of up to ``world`` consecutive frames (``n_valid`` of them exist).  ``frame``/``depth_local`` are
        this rank's frame and its depth (None if rank >= n_valid).  Pass ``depth_all`` [world,h,w(,3)] when every
        rank already has all depth planes (precomputed depth video) to skip the collective.  Returns this rank's
        muxed frame or None."""
        G = self.world
        if depth_all is None:
            if G == 1:
                depth_all = depth_local[None]
            else:
                assert depth_local is not None, "every rank contributes a (possibly dummy) depth plane to the all-gather"
                shp = tuple(depth_local.shape)
                flat = torch.empty((G * shp[0],) + shp[1:], dtype=depth_local.dtype, device=depth_local.device)
                dist.all_gather_into_tensor(flat, depth_local.contiguous(), group=self.group)  # concatenation along dim 0
                depth_all = flat.view((G,) + shp)
        out = None
        for g in range(min(n_valid, G)):
            if g == self.rank:
                out = self.b.render_frame(frame, depth_all[g])
            else:
                self.b.advance_state(depth_all[g])
        return out

    def render_clip(self, n_frames: int, get_frame: Callable[[int], torch.Tensor],
                    get_depth: Callable[[int], torch.Tensor], depth_everywhere: bool = False) -> Iterable:
        """Render frames 0..n_frames-1 of a clip (the caller has already dropped the reference's skipped first frame).
        ``get_frame(t)`` / ``get_depth(t)`` are called only for frames this rank needs.  Yields (t, muxed frame) for
        the frames this rank owns."""
        G = self.world
        self.new_clip()
        for r0 in range(0, n_frames, G):
            n_valid = min(G, n_frames - r0)
            t = r0 + self.rank
            mine = self.rank < n_valid
            frame = get_frame(t) if mine else None
            if depth_everywhere:
                d_all = torch.stack([get_depth(r0 + g) for g in range(n_valid)])
                out = self.render_round(n_valid, frame, None, d_all)
            else:
                d_loc = get_depth(t) if mine else torch.zeros_like(get_depth(r0))  # dummy contribution past the clip end
                out = self.render_round(n_valid, frame, d_loc)
            if mine:
                yield t, out


class StepShardedRenderer:
    """Three-phase frame sharding (include/vd3d.h "frame sharding"): the scaling path used by bench.py for N > 1.

    A step is a window of ``world * B`` consecutive frames; global order inside the step is t = j * world + g
    (j-th local frame of rank g), i.e. frames are dealt round-robin.  Per step and rank:

      0. (caller) depth inference for the B local frames, all-gather of the uint8 depth planes  -> depth_all [world*B,h,w]
      1. pass 1 over ALL world*B frames in order: own frames -> full measurement (warp-res select, s1) with planes kept
         in a slot; foreign frames -> eye-res chain only (~5x cheaper than a full state advance)
      2. all-gather of the measured s1 (B floats per rank), tracker replay over the step (one tiny kernel)
      3. pixel pass (shift plane, fused warp, fused finish) for the B own frames

    Every rank runs the identical eye-res chain and the identical replay, so all tracker state is bit-identical across
    ranks and the muxed frames equal the sequential 1-GPU render (tests/test_hip_parity.py emulates two ranks with two
    contexts on one GPU and compares bit for bit).
    """

    def __init__(self, renderer, params, rank: int, world: int, frames_per_rank: int, group=None):
        self.r, self.p, self.rank, self.world, self.B, self.group = renderer, params, rank, world, frames_per_rank, group
        if world * frames_per_rank > 512:
            raise ValueError("a sharded step holds at most 512 frames")
        renderer.shard_begin(params, frames_per_rank)
        self.s1_local = torch.zeros(frames_per_rank, dtype=torch.float32, device=renderer.device)
        self.own_slots = [(t // world if t % world == rank else -1) for t in range(world * frames_per_rank)]

    def gather(self, local: torch.Tensor) -> torch.Tensor:
        """all-gather along dim 0: [B, ...] per rank -> [world*B, ...] laid out rank-major."""
        if self.world == 1:
            return local
        shp = tuple(local.shape)
        out = torch.empty((self.world * shp[0],) + shp[1:], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def pass1(self, frames_local, depth_all: torch.Tensor):
        G, B = self.world, self.B
        for j in range(B):
            for g in range(G):
                t = j * G + g
                d = depth_all[g * B + j]
                if g == self.rank:
                    self.r.shard_pass1(frames_local[j], d, self.p, t, slot=j, s1_out=self.s1_local[j:j + 1])
                else:
                    self.r.shard_pass1(None, d, self.p, t, slot=-1)

    def finish(self, s1_gathered: torch.Tensor, outs=None):
        G, B = self.world, self.B
        s1_t = s1_gathered.view(G, B).t().contiguous().view(-1) if G > 1 else s1_gathered   # rank-major -> frame order
        self.r.shard_pass2(s1_t, self.own_slots, self.p)
        res = []
        for j in range(B):
            res.append(self.r.shard_pixels(j, self.p, out=None if outs is None else outs[j]))
        return res

    def render_step(self, frames_local, depth_local: torch.Tensor, outs=None):
        depth_all = self.gather(depth_local)
        self.pass1(frames_local, depth_all)
        return self.finish(self.gather(self.s1_local), outs)


class MeasureReplaySharder:
    """Measure / replay frame sharding (include/vd3d.h ``vd3d_shard2_*``): the N > 1 path of bench.py.

    Same step layout as StepShardedRenderer (global order t = j * world + g), but the only REPLICATED work per foreign
    frame is the TemporalDepthFilter plane EMA (one small launch).  Per step and rank:

      0. (caller) depth inference for the B local frames, all-gather of the uint8 depth planes       [collective 1]
      1. P1 over all world*B frames in order: foreign -> plane EMA; own -> ingest + plane EMA + exact q.02/q.98
      2. all-gather of (q_lo, q_hi) per frame (8 B x frames)                                          [collective 2]
         R1: DepthPercentileEMA replayed on every rank
      3. P3 per own frame: normalise, eye-res statistics, warp-res select chain, shaped depth, s1
      4. all-gather of 4 x int64 per frame                                                            [collective 3]
         R2: dynamic parallax scale, ShiftSmoother, FocalDepthTracker, FloatingWindowTracker, ConvergenceEMA,
         FloatingBarEaser replayed in frame order on every rank; own slots patched
      5. pixel pass (shift plane, fused warp, fused finish) per own frame

    Output equals the sequential 1-GPU render bit for bit (tests/test_hip_parity.py, ranks emulated with one context each).
    """

    def __init__(self, renderer, params, rank: int, world: int, frames_per_rank: int, group=None, slot_base: int = 0):
        """``slot_base``: first slot of the renderer this sharder uses (two sharders with slot_base 0 and B on one renderer keep
        two steps in flight: the pixel pass of step i-1 can run after the measurements of step i were enqueued, bench.py)."""
        self.r, self.p, self.rank, self.world, self.B, self.group = renderer, params, rank, world, frames_per_rank, group
        self.slot_base = int(slot_base)
        if world * frames_per_rank > 512:
            raise ValueError("a sharded step holds at most 512 frames")
        renderer.shard_begin(params, self.slot_base + frames_per_rank)
        self.q_local = torch.zeros((frames_per_rank, 2), dtype=torch.float32, device=renderer.device)
        self.m_local = torch.zeros((frames_per_rank, 4), dtype=torch.int64, device=renderer.device)
        self.c_local = torch.zeros((frames_per_rank, 4), dtype=torch.int32, device=renderer.device)
        self.auto_crop = bool(getattr(params, "auto_crop_black_bars", 0)) if params is not None else False
        self.own_slots = [(self.slot_base + t // world if t % world == rank else -1) for t in range(world * frames_per_rank)]

    def gather(self, local: torch.Tensor) -> torch.Tensor:
        """all-gather along dim 0: [B, ...] per rank -> [world*B, ...] laid out rank-major."""
        if self.world == 1:
            return local
        shp = tuple(local.shape)
        out = torch.empty((self.world * shp[0],) + shp[1:], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def _frame_order(self, gathered: torch.Tensor) -> torch.Tensor:
        """rank-major [world*B, k] -> frame order t = j * world + g."""
        if self.world == 1:
            return gathered
        G, B = self.world, self.B
        return gathered.view(G, B, -1).transpose(0, 1).reshape(G * B, -1).contiguous()

    # ``n_valid`` (all methods): number of real frames in this step (default: world * B).  The LAST step of a clip may be partial:
    # frames t >= n_valid do not exist, their owners skip them, the collectives still move full-size buffers, and the replays stop
    # at n_valid.  Valid frames are always a prefix of the step's frame order.
    def _nv(self, n_valid):
        n = self.world * self.B if n_valid is None else int(n_valid)
        if not (1 <= n <= self.world * self.B):
            raise ValueError("n_valid out of range")
        return n

    def p0(self, frames_local, n_valid=None):
        """auto_crop_black_bars: black-bar detection on the own frames, all-gather of the crop rectangles (16 B x frames)."""
        n = self._nv(n_valid)
        for j in range(self.B):
            if j * self.world + self.rank < n:
                self.r.shard2_p0(frames_local[j], self.p, self.c_local[j])
        self.r.shard2_set_crops(self._frame_order(self.gather(self.c_local))[:n])

    def p1(self, frames_local, depth_all: torch.Tensor, n_valid=None):
        """Every frame of the step in order; runs of consecutive foreign frames go down as ONE launch each (per-pixel EMA chains
        are independent), so a rank issues about 2 * B launches for the foreign frames of a step instead of (world-1) * B."""
        G, B = self.world, self.B
        n = self._nv(n_valid)
        run, run_first = [], 0
        multi = hasattr(self.r, "shard2_p1_foreign")

        def flush():
            nonlocal run
            if run:
                if multi:
                    self.r.shard2_p1_foreign(run, self.p, run_first)
                else:
                    for k, d in enumerate(run):
                        self.r.shard2_p1(None, d, self.p, run_first + k, slot=-1)
                run = []

        for j in range(B):
            for g in range(G):
                t = j * G + g
                if t >= n:
                    continue
                d = depth_all[g * B + j]
                if g == self.rank:
                    flush()
                    self.r.shard2_p1(frames_local[j], d, self.p, t, slot=self.slot_base + j, q_out=self.q_local[j])
                else:
                    if not run:
                        run_first = t
                    run.append(d)
        flush()

    def p3(self, n_valid=None):
        n = self._nv(n_valid)
        for j in range(self.B):
            if j * self.world + self.rank < n:
                self.r.shard2_p3(self.slot_base + j, j * self.world + self.rank, self.p, self.m_local[j])

    def finish(self, m_gathered: torch.Tensor, outs=None, ordered: bool = False, n_valid=None):
        """``ordered``: m_gathered is already in frame order (callers that run the renderer on a private stream do the reordering
        inside that stream's context, see bench.py).  Returns the muxed frames of the valid own frames."""
        n = self._nv(n_valid)
        m = m_gathered if ordered else self._frame_order(m_gathered)
        self.r.shard2_r2(m[:n], self.own_slots[:n], self.p)
        return self.pixels(outs, n_valid)

    def replay(self, m_ordered: torch.Tensor, n_valid=None):
        """R2 only (frame-ordered records): the pixel pass can then be issued later with ``pixels``."""
        n = self._nv(n_valid)
        self.r.shard2_r2(m_ordered[:n], self.own_slots[:n], self.p)

    def pixels(self, outs=None, n_valid=None):
        n = self._nv(n_valid)
        return [self.r.shard_pixels(self.slot_base + j, self.p, out=None if outs is None else outs[j]) for j in range(self.B)
                if j * self.world + self.rank < n]

    def render_step(self, frames_local, depth_local: torch.Tensor, outs=None, n_valid=None):
        if self.auto_crop:
            self.p0(frames_local, n_valid)
        self.p1(frames_local, self.gather(depth_local), n_valid)
        self.r.shard2_r1(self._frame_order(self.gather(self.q_local))[:self._nv(n_valid)])
        self.p3(n_valid)
        return self.finish(self.gather(self.m_local), outs, n_valid=n_valid)

    def render_clip(self, n_frames: int, get_frame, get_depth, overlap_pixels: bool = False):
        """Render frames 0..n_frames-1 of one clip (callables return device tensors: uint8 BGR frame / depth plane of frame t).
        Yields (t, muxed_frame) for the frames this rank owns, in increasing t; the last step may be partial.  Every rank must
        call this with the same n_frames (the collectives are matched).

        ``overlap_pixels``: steps alternate between two slot sets and the pixel kernels of step i run on the renderer's second
        stream (``vd3d_set_pixel_overlap``) while the measurement chain of step i+1 runs on the first; the frames of step i are
        yielded once step i+1 has been enqueued and their pixel passes have completed (host wait per frame).  Same bytes."""
        G, B = self.world, self.B
        per_step = G * B
        proto_f, proto_d = None, None
        sets = [self]
        if overlap_pixels:
            if getattr(self, "_twin", None) is None:
                self._twin = MeasureReplaySharder(self.r, self.p, self.rank, G, B, self.group, slot_base=self.slot_base + B)
            sets = [self, self._twin]
            self.r.set_pixel_overlap(True)

        def drain(step):
            own, outs, sh = step
            for j, (t, o) in enumerate(zip(own, outs)):
                self.r.wait_pixels(sh.slot_base + j)
                yield t, o

        pending = None
        try:
            for k, base in enumerate(range(0, n_frames, per_step)):
                nv = min(per_step, n_frames - base)
                fl, dl = [], []
                for j in range(B):
                    t = base + j * G + self.rank
                    if t < n_frames:
                        f, d = get_frame(t), get_depth(t)
                        proto_f, proto_d = f, d
                    else:   # past the end of the clip: a dummy contribution keeps the collectives full-size
                        if proto_f is None:
                            proto_f, proto_d = get_frame(n_frames - 1), get_depth(n_frames - 1)
                        f, d = torch.zeros_like(proto_f), torch.zeros_like(proto_d)
                    fl.append(f); dl.append(d)
                sh = sets[k % len(sets)]
                outs = sh.render_step(fl, torch.stack(dl), n_valid=nv)
                own = [base + j * G + self.rank for j in range(B) if j * G + self.rank < nv]
                if overlap_pixels:
                    if pending is not None:
                        yield from drain(pending)
                    pending = (own, outs, sh)
                else:
                    for t, o in zip(own, outs):
                        yield t, o
            if pending is not None:
                yield from drain(pending)
        finally:
            if overlap_pixels:
                self.r.set_pixel_overlap(False)   # joins whatever is still in flight
