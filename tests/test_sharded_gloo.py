"""Multi-process CPU tests (gloo, world_size 2, 3 and -- round 5 -- 8) of the chunked frame-sharding protocol (visiondepth3d_amd.sharded.ChunkSharder)
with the CPU ORACLE as backend (tests/oracle_chunk.py): the real point-to-point plane hand-off, the two all-gathers and the final
broadcast run over gloo, every stage computes real numbers, and the sharded clip must equal the sequential oracle render BIT FOR
BIT -- muxed frames, final tracker state and final plane state on every rank.  The clip has a partial last step
(one rank with a short chunk, with world 3 also ranks that only forward the plane) and a skip_blank_frames hit.  The same
orchestrator drives the HIP backend over RCCL on the GPU box (bench.py --gpus N).  World 8 (the size of the driver's SCALE run): 19 frames in
chunks of 2 (one full step of 16 + a last step in which rank 0 owns two frames, rank 1 one and ranks 2 .. 7 only forward the plane) and 11
frames in chunks of 1, each with blank frames and collapsing depth frames in BOTH steps."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visiondepth3d_amd import synth
from visiondepth3d_amd.params import render_kwargs_to_params

SH, SW, NF, B = 72, 128, 7, 2
BLANK = {2}
KW = dict(output_format="Half-SBS", output_height=72, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
          dof_strength=2.0, feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True,
          skip_blank_frames=True, ipd_factor=1.1, aten_sum_threads=3)   # torch.mean in ATen's summation order with 3 threads (round 5): part of the replayed record
STATE_FIELDS = ("fw_prev_offset", "fw_frame_counter", "ema_valid", "ema_lo", "ema_hi", "conv_valid", "bar_prev_width", "conv_val",
                "focal_valid", "smooth_valid", "focal", "sm_fg", "sm_mg", "sm_bg", "tdf_valid", "prev_depth_valid")


def _blank(nf):
    return BLANK | ({nf - 2} if nf > 8 else set())                     # longer clips: a blank frame in the last (partial) step too


def _clip(nf=NF):
    frames, depths = synth.synth_clip(nf, SH, SW)
    gray = [synth.depth_to_u8_bgr(d)[..., 0].copy() for d in depths]   # uint8 [h,w] depth planes
    gray[4][:] = 90                                                    # a collapsing depth frame (DepthPercentileEMA guard)
    if nf > 8:
        gray[nf - 1][:] = 31                                           # ... and one as the clip's last frame (a short chunk of the last step)
    return frames, gray


def _state_vec(st):
    return np.array([float(getattr(st, k)) for k in STATE_FIELDS], np.float64)


def _sequential(oracle, nf=NF):
    frames, gray = _clip(nf)
    p = render_kwargs_to_params(SW, SH, **KW)
    ro = oracle.RenderOracle(p)
    ro.new_clip()
    blank = _blank(nf)
    seq = [ro.render(f, g, 2, blank=(t in blank)) for t, (f, g) in enumerate(zip(frames, gray))]
    return seq, _state_vec(ro.state), ro.tdf_prev.copy()


def _worker(rank, world, port, outdir, nf=NF, b=B):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_chunk import OracleChunkBackend
        from visiondepth3d_amd.sharded import ChunkSharder
        torch.set_num_threads(1)
        frames, gray = _clip(nf)
        p = render_kwargs_to_params(SW, SH, **KW)
        be = OracleChunkBackend(p)
        sr = ChunkSharder(be, rank, world, b)
        got = {}
        for t, out in sr.render_clip(nf, lambda t: torch.from_numpy(frames[t]), lambda t: torch.from_numpy(gray[t]), blank_frames=_blank(nf)):
            got[t] = out.numpy()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **{str(k): v for k, v in got.items()})
        np.save(os.path.join(outdir, f"state{rank}.npy"), _state_vec(be.state))
        np.save(os.path.join(outdir, f"plane{rank}.npy"), be.tdf)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,nf,b", [(2, NF, B), (3, NF, B), (8, 19, 2), (8, 11, 1)])
def test_chunk_sharding_over_gloo_equals_sequential(tmp_path, oracle, world, nf, b):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), nf, b), nprocs=world, join=True)
    seq, st_seq, plane_seq = _sequential(oracle, nf)
    owned = {}
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for k in z.files:
            t = int(k)
            assert (t % (world * b)) // b == r          # contiguous chunks: frame t of a step belongs to rank t // B
            owned[t] = z[k]
    if world == 8:   # the last step is partial: ranks that own nothing in it only forward the plane, and still end in the sequential state (below)
        last = {(t % (world * b)) // b for t in range((nf // (world * b)) * world * b, nf)}
        assert 0 < len(last) < world
    assert sorted(owned) == list(range(nf))
    for t in range(nf):
        assert np.array_equal(owned[t], seq[t]), t
    for r in range(world):   # every rank ends in the sequential render's tracker AND plane state
        assert np.array_equal(np.load(tmp_path / f"state{r}.npy"), st_seq), r
        assert np.array_equal(np.load(tmp_path / f"plane{r}.npy"), plane_seq), r


@pytest.mark.parametrize("G", [1, 2, 4])
def test_chunk_sharding_emulated_in_process(oracle, G):
    """The same protocol with the exchanges done by hand (tests/shard_emul.py): fast coverage of more world sizes, incl. world 1
    (degenerates to the sequential render) and world 4 (7 frames, B = 2: two ranks never own a frame of the last step)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_chunk import OracleChunkBackend
    from shard_emul import Emu
    from visiondepth3d_amd.sharded import ChunkSharder
    frames, gray = _clip()
    p = render_kwargs_to_params(SW, SH, **KW)
    bes = [OracleChunkBackend(p) for _ in range(G)]
    for be in bes:
        be.new_clip()
    emu = Emu([ChunkSharder(be, g, G, B) for g, be in enumerate(bes)])
    got = emu.run_clip([torch.from_numpy(f) for f in frames], [torch.from_numpy(g) for g in gray], B, blank_frames=BLANK)
    seq, st_seq, plane_seq = _sequential(oracle)
    for t in range(NF):
        assert np.array_equal(got[t].numpy(), seq[t]), t
    for g, be in enumerate(bes):
        assert np.array_equal(_state_vec(be.state), st_seq)
        assert np.array_equal(be.tdf, plane_seq)
        # the sharder drove the BATCHED stage entry points (what HipChunkBackend offers: one call per stage and step for the consecutive own
        # frames), with this rank's first step index and the slot set's first slot; partial last steps shorten the batch
        assert be.batch_calls and {c[0] for c in be.batch_calls} == {"p1", "p3"}
        assert all(1 <= n <= B and idx0 == g * B and slot0 == 0 for (_, n, idx0, slot0) in be.batch_calls), be.batch_calls


def test_protocol_traffic_is_records_plus_one_plane_per_boundary():
    """SURVEY 8(e) / north_star: the data path exchanges scalars and ONE filtered plane per chunk boundary -- never depth planes
    of all frames.  Counted on the orchestrator with a byte-counting stand-in for torch.distributed."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from visiondepth3d_amd import sharded

    class Be:
        device = torch.device("cpu")
        def begin(self, n): pass
        def new_clip(self): pass
        def plane_shape(self): return (540, 960)
        def plane_export(self, out=None): return torch.zeros(540, 960)
        def plane_import(self, plane, valid=True): pass
        def p1(self, *a): pass
        def r1(self, q): pass
        def p3(self, *a): pass
        def r2(self, *a): pass
        def pixels(self, slot, out=None, blank_frame=None): return torch.zeros(1)

    sent = {"p2p": 0, "gather": 0}

    class FakeDist:
        @staticmethod
        def send(t, dst, group=None): sent["p2p"] += t.numel() * t.element_size()
        @staticmethod
        def recv(t, src, group=None): pass
        @staticmethod
        def all_gather_into_tensor(out, inp, group=None): sent["gather"] += inp.numel() * inp.element_size()
    real = sharded.dist
    sharded.dist = FakeDist
    try:
        world, Bf = 8, 16
        sr = sharded.ChunkSharder(Be(), 3, world, Bf)
        sr.render_step([None] * Bf, [None] * Bf, first_step=True, more_steps=True)
    finally:
        sharded.dist = real
    assert sent["p2p"] == 540 * 960 * 4                      # one eye-size float32 plane to the next chunk's owner (2.1 MB @1080p)
    assert sent["gather"] == Bf * (2 * 4 + 4 * 8)            # 40 B per own frame of records
