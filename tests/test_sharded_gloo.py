"""world_size-2 gloo test (CPU) of the frame-sharded runner: orchestration, the depth all-gather and the
state-advance protocol, with the CPU oracle as backend.  The sharded result must equal the sequential render
bit for bit.  The same runner drives the HIP backend over RCCL on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visiondepth3d_amd import synth
from visiondepth3d_amd.params import render_kwargs_to_params
from visiondepth3d_amd.sharded import FrameShardedRenderer

SH, SW, NF = 72, 128, 7   # odd frame count: the last round is partial
KW = dict(output_format="Half-SBS", output_height=72, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
          dof_strength=2.0, feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)


class OracleBackend:
    def __init__(self, params):
        from oracle import oracle as O
        self.ro = O.RenderOracle(params)
        self.p = params
        self.zero = np.zeros((params.src_h, params.src_w, 3), np.uint8)

    def new_clip(self):
        self.ro.new_clip()

    def render_frame(self, frame, depth):
        return torch.from_numpy(self.ro.render(frame.numpy(), depth.numpy(), 2 if depth.dim() == 2 else 1))

    def advance_state(self, depth):  # the state does not depend on the frame's pixels
        self.ro.render(self.zero, depth.numpy(), 2 if depth.dim() == 2 else 1)


def _clip():
    frames, depths = synth.synth_clip(NF, SH, SW)
    gray = [synth.depth_to_u8_bgr(d)[..., 0].copy() for d in depths]   # uint8 [h,w] depth planes
    return frames, gray


def _worker(rank, world, port, outdir, depth_everywhere):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames, gray = _clip()
        p = render_kwargs_to_params(SW, SH, **KW)
        sr = FrameShardedRenderer(OracleBackend(p), rank, world)
        got = {}
        for t, out in sr.render_clip(NF, lambda t: torch.from_numpy(frames[t]), lambda t: torch.from_numpy(gray[t]),
                                     depth_everywhere=depth_everywhere):
            got[t] = out.numpy()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **{str(k): v for k, v in got.items()})
        st = sr.b.ro.state
        np.save(os.path.join(outdir, f"state{rank}.npy"), np.array([st.fw_prev_offset, st.ema_lo, st.ema_hi, st.conv_val, st.focal]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("depth_everywhere", [False, True])
def test_frame_sharding_world2_equals_sequential(tmp_path, oracle, depth_everywhere):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), depth_everywhere), nprocs=world, join=True)
    # sequential reference
    frames, gray = _clip()
    p = render_kwargs_to_params(SW, SH, **KW)
    ro = oracle.RenderOracle(p)
    ro.new_clip()
    seq = [ro.render(f, g, 2) for f, g in zip(frames, gray)]
    owned = {}
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for k in z.files:
            assert int(k) % world == r
            owned[int(k)] = z[k]
    assert sorted(owned) == list(range(NF))
    for t in range(NF):
        assert np.array_equal(owned[t], seq[t]), t
    # every rank walked the identical state trajectory
    s0, s1 = np.load(tmp_path / "state0.npy"), np.load(tmp_path / "state1.npy")
    assert np.array_equal(s0, s1)
    assert s0[0] == ro.state.fw_prev_offset and s0[4] == ro.state.focal


def test_single_rank_degenerates_to_sequential(oracle):
    frames, gray = _clip()
    p = render_kwargs_to_params(SW, SH, **KW)
    sr = FrameShardedRenderer(OracleBackend(p), 0, 1)
    got = [o.numpy() for _, o in sr.render_clip(NF, lambda t: torch.from_numpy(frames[t]), lambda t: torch.from_numpy(gray[t]))]
    ro = oracle.RenderOracle(p)
    ro.new_clip()
    for t in range(NF):
        assert np.array_equal(got[t], ro.render(frames[t], gray[t], 2))


# ---------------------------------------------------------------------------------------------------
# three-phase protocol (StepShardedRenderer): orchestration + collectives on gloo with a recording fake backend
# ---------------------------------------------------------------------------------------------------
class _FakeRenderer:
    """Implements the shard_* surface of Renderer on CPU tensors and records what the protocol asked of it."""
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []
        self.replayed = None

    def shard_begin(self, params, n_slots):
        self.n_slots = n_slots

    def shard_pass1(self, frame, depth, params, step_idx, slot=-1, s1_out=None):
        key = int(depth[0, 0])            # the depth plane encodes the global frame index
        self.calls.append((step_idx, slot, key, frame is not None))
        if slot >= 0:
            assert frame is not None and int(frame[0, 0, 0]) == key
            s1_out[0] = 1000.0 + key      # "measurement" of the owned frame

    def shard_pass2(self, s1_all, own_slots, params):
        self.replayed = (s1_all.clone(), list(own_slots))

    def shard_pixels(self, slot, params, out=None):
        return torch.tensor([slot])


def _proto_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from visiondepth3d_amd.sharded import StepShardedRenderer
        B = 3
        fr = _FakeRenderer()
        sr = StepShardedRenderer(fr, None, rank, world, B)
        # global frame t = j*world + g carries the value t in its depth plane and frame
        depth_local = torch.stack([torch.full((4, 5), j * world + rank, dtype=torch.uint8) for j in range(B)])
        frames_local = [torch.full((4, 5, 3), j * world + rank, dtype=torch.uint8) for j in range(B)]
        outs = sr.render_step(frames_local, depth_local)
        s1_all, own = fr.replayed
        torch.save({"calls": fr.calls, "s1": s1_all, "own": own, "outs": [int(o) for o in outs]}, os.path.join(outdir, f"p{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_three_phase_protocol_world2(tmp_path):
    world, B = 2, 3
    mp.spawn(_proto_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        rec = torch.load(tmp_path / f"p{rank}.pt")
        # pass 1 visited every frame of the step exactly once, in frame order, owning exactly its round-robin share
        assert [c[0] for c in rec["calls"]] == list(range(world * B))
        assert [c[2] for c in rec["calls"]] == list(range(world * B))          # depth plane of frame t really is frame t's
        for t, (idx, slot, key, has_frame) in enumerate(rec["calls"]):
            assert (slot >= 0) == (t % world == rank) == has_frame
            if slot >= 0:
                assert slot == t // world
        # the replay saw every frame's s1 in FRAME order on every rank (all-gather is rank-major: needs the transpose)
        assert rec["s1"].tolist() == [1000.0 + t for t in range(world * B)]
        assert rec["own"] == [(t // world if t % world == rank else -1) for t in range(world * B)]
        assert rec["outs"] == list(range(B))


# ---------------------------------------------------------------------------------------------------
# measure / replay protocol (MeasureReplaySharder, the N > 1 path of bench.py): orchestration + the three all-gathers on gloo
# ---------------------------------------------------------------------------------------------------
class _FakeRenderer2:
    """shard2_* surface of Renderer on CPU tensors; records the order of calls and what the replays were given."""
    device = torch.device("cpu")

    def __init__(self):
        self.log, self.r1, self.r2 = [], None, None

    def shard_begin(self, params, n_slots):
        self.n_slots = n_slots

    def shard2_p1(self, frame, depth, params, step_idx, slot=-1, q_out=None):
        key = int(depth[0, 0])
        self.log.append(("p1", step_idx, slot, key, frame is not None))
        if slot >= 0:
            assert int(frame[0, 0, 0]) == key
            q_out[0] = 10.0 + key; q_out[1] = 20.0 + key

    def shard2_r1(self, q_all):
        self.log.append(("r1",))
        self.r1 = q_all.clone()

    def shard2_p3(self, slot, step_idx, params, m_out):
        self.log.append(("p3", step_idx, slot))
        m_out[:] = torch.tensor([step_idx, 100 + step_idx, 200 + step_idx, 300 + step_idx], dtype=torch.int64)

    def shard2_r2(self, m_all, own_slots, params):
        self.log.append(("r2",))
        self.r2 = (m_all.clone(), list(own_slots))

    def shard_pixels(self, slot, params, out=None):
        self.log.append(("px", slot))
        return torch.tensor([slot])

    def set_pixel_overlap(self, on):
        self.log.append(("ov", bool(on)))

    def wait_pixels(self, slot):
        self.log.append(("wait", slot))


def _proto2_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from visiondepth3d_amd.sharded import MeasureReplaySharder
        B = 3
        fr = _FakeRenderer2()
        sr = MeasureReplaySharder(fr, None, rank, world, B)
        depth_local = torch.stack([torch.full((4, 5), j * world + rank, dtype=torch.uint8) for j in range(B)])
        frames_local = [torch.full((4, 5, 3), j * world + rank, dtype=torch.uint8) for j in range(B)]
        outs = sr.render_step(frames_local, depth_local)
        rec = {"log": list(fr.log), "r1": fr.r1, "r2": fr.r2, "outs": [int(o) for o in outs]}
        # a whole clip whose length (8) is not a multiple of world * B (6): one full step + a partial one
        fr.log.clear()
        got = list(sr.render_clip(8, lambda t: torch.full((4, 5, 3), t % 6, dtype=torch.uint8), lambda t: torch.full((4, 5), t % 6, dtype=torch.uint8)))
        rec["clip_owned"] = [t for t, _ in got]
        rec["clip_log_kinds"] = [e[0] for e in fr.log]
        rec["clip_r2_len"] = int(fr.r2[0].shape[0])
        # the same clip (14 frames: two full steps + a partial one) with overlapped pixel passes
        fr.log.clear()
        got = list(sr.render_clip(14, lambda t: torch.full((4, 5, 3), t % 6, dtype=torch.uint8), lambda t: torch.full((4, 5), t % 6, dtype=torch.uint8),
                                  overlap_pixels=True))
        rec["ov_owned"] = [t for t, _ in got]
        rec["ov_log"] = [e for e in fr.log if e[0] in ("ov", "px", "wait", "r2")]
        torch.save(rec, os.path.join(outdir, f"q{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_measure_replay_protocol_world2(tmp_path):
    world, B = 2, 3
    mp.spawn(_proto2_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    n = world * B
    for rank in range(world):
        rec = torch.load(tmp_path / f"q{rank}.pt")
        log = rec["log"]
        p1 = [e for e in log if e[0] == "p1"]
        # P1 visits every frame of the step once, in frame order, with frame t's depth plane; owns its round-robin share
        assert [e[1] for e in p1] == list(range(n)) and [e[3] for e in p1] == list(range(n))
        for t, (_, idx, slot, key, has_frame) in enumerate(p1):
            assert (slot >= 0) == (t % world == rank) == has_frame and (slot < 0 or slot == t // world)
        # phase order: all P1, then R1, then the own P3s, then R2, then the pixel passes
        kinds = [e[0] for e in log]
        assert kinds == ["p1"] * n + ["r1"] + ["p3"] * B + ["r2"] + ["px"] * B
        assert [e[1] for e in log if e[0] == "p3"] == [j * world + rank for j in range(B)]
        # both replays see every frame's record in FRAME order (the all-gather is rank-major: needs the transpose)
        assert rec["r1"].tolist() == [[10.0 + t, 20.0 + t] for t in range(n)]
        m_all, own = rec["r2"]
        assert m_all.tolist() == [[t, 100 + t, 200 + t, 300 + t] for t in range(n)]
        assert own == [(t // world if t % world == rank else -1) for t in range(n)]
        assert rec["outs"] == list(range(B))
        # render_clip: 8 frames = step of 6 + partial step of 2 (frames 6, 7 -> one own frame per rank); replays stop at n_valid
        assert rec["clip_owned"] == [t for t in range(8) if t % world == rank]
        assert rec["clip_log_kinds"] == ["p1"] * 6 + ["r1"] + ["p3"] * 3 + ["r2"] + ["px"] * 3 + ["p1"] * 2 + ["r1"] + ["p3"] + ["r2"] + ["px"]
        assert rec["clip_r2_len"] == 2
        # overlapped: frames still come out in order; steps alternate between slot sets {0,1,2} and {3,4,5}; a step's frames are
        # waited for (and yielded) only after the NEXT step's chain and pixel passes were enqueued; overlap is switched off at the end
        assert rec["ov_owned"] == [t for t in range(14) if t % world == rank]
        px = lambda sl: [("px", s_) for s_ in sl]
        wt = lambda sl: [("wait", s_) for s_ in sl]
        assert rec["ov_log"] == ([("ov", True), ("r2",)] + px([0, 1, 2]) + [("r2",)] + px([3, 4, 5]) + wt([0, 1, 2]) +
                                 [("r2",)] + px([0]) + wt([3, 4, 5]) + wt([0]) + [("ov", False)])
