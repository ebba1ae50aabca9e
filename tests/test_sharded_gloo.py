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
