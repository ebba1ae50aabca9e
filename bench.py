#!/usr/bin/env python3
"""bench.py -- stereo-pairs/s of the depth -> DIBR hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--batch B] [--depth-dtype f32|bf16]

A "step" is one pass of the hot path over one batch of B synthetic frames already resident in HBM:
depth inference (workloads with a depth net, float32 like the reference's Hugging Face path) -> 8-bit depth hand-off ->
the per-frame DIBR chain (ingest, exact order statistics, shaping, warp + feather, DOF, grade, sharpen, Half-SBS mux) ->
muxed frames in HBM.

Default run (no --workload), N = 1: the HEADLINE is BASELINE.json configs[3]'s per-GPU slice `4k-dav2b-dibr` (3840x2160,
Depth-Anything-V2-Base + full DIBR: the configuration every `north_star` target is quoted on), timed over exactly --steps
steps after --warmup steps; the same invocation then measures, as labelled sub-records of the one JSON line,
  * `4k-dibr`          configs[2], the HBM-roofline run (>= 200 timed frames, DIBR only, W1 / E1 alone on the GPU),
  * `1080p-dav2s-dibr` configs[1] (1080p, DA-V2-Small + DIBR),
  * `1080p-dibr`       the DIBR-only GPU rate that does the SAME work as `cpu_baseline_1080p` (the C oracle has no depth net),
  * the bf16 variant of the headline (reduced precision, labelled; never `value`).
For N > 1 the driver launches one rank per GPU with torch.distributed.run; the frames of ONE clip are sharded in contiguous
chunks (visiondepth3d_amd/sharded.py) and only the headline workload runs.  Weak scaling: B frames per rank per step.  The
timed region is bracketed by barrier + synchronize and the MAX over ranks is reported.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (src_h, src_w, depth model or None, description)
    "1080p-dav2s-dibr": (1080, 1920, "depth-anything-v2-small", "BASELINE configs[1]: 1080p, DA-V2-Small inference + DIBR, Half-SBS"),
    "1080p-dibr": (1080, 1920, None, "1080p, precomputed f32 depth, DIBR only, Half-SBS"),
    "4k-dibr": (2160, 3840, None, "BASELINE configs[2]: 4K, precomputed f32 depth, DIBR warp+DOF only (HBM roofline run)"),
    "4k-dav2b-dibr": (2160, 3840, "depth-anything-v2-base", "BASELINE configs[3] per-GPU slice: 4K, DA-V2-Base + DIBR"),
    "4k-dibr-sepdof": (2160, 3840, None, "4K DIBR only with dof_dense_conv=0: separable DOF levels instead of the reference's dense k x k "
                       "convolution order (faster finishing kernel; differs from the reference on ~0.5 % of samples, max 4)"),
    "4k-dibr-dof3": (2160, 3840, None, "4K DIBR only with dof_strength 3.0 (13-tap Gaussian: beyond the fused finishing kernel's 9 taps, the unfused "
                     "DOF + sharpen / mux kernels run)"),
    "4k-dibr-anaglyph": (2160, 3840, None, "4K DIBR only, Red-Cyan Anaglyph output (fused finishing kernel since round 4)"),
    "4k-dibr-vr": (2160, 3840, None, "4K DIBR only, VR output (1440x1600 canvas per eye: a fractional 8/3 INTER_AREA fit + letterbox, which the fused "
                   "finishing kernel does not take: the unfused DOF + sharpen / fit / mux kernels run)"),
    "1080p-gui-defaults": (1080, 1920, None, "1080p DIBR only with the GUI's OWN default configuration (VisionDepth3D.py:1405-1453: Full-SBS, fg 4.5 / mg -1.5 / "
                           "bg -6, blur_ksize 1, feather_strength 0.0, sharpness 0.2, zero_parallax 0.01): feather_shift_edges is an exact no-op there, so W1 "
                           "runs without the mask kernel, the window sums and the blend (round 5)"),
    "4k-dibr-gui": (2160, 3840, None, "4K DIBR only with the GUI's default configuration (Full-SBS 7680x2160 output, feather_strength 0.0: the exact "
                    "no-feather W1) -- the configuration in which north_star's HBM question about the warp kernel is meaningful"),
    "4k-dibr-gui-hsbs": (2160, 3840, None, "4K DIBR only, GUI default controls with Half-SBS output (the 2:1 eye resize inside W1, no feathering)"),
}
GUI_KW = dict(output_format="Full-SBS", fg_shift=4.5, mg_shift=-1.5, bg_shift=-6.0, sharpness_factor=0.2, dof_strength=2.0, feather_strength=0.0,
              blur_ksize=1, use_subject_tracking=True, use_floating_window=True, zero_parallax_strength=0.01)   # VisionDepth3D.py:1405-1453
# render keyword overrides of the variants above (everything else: RENDER_KW)
WORKLOAD_KW = {"4k-dibr-dof3": dict(dof_strength=3.0), "4k-dibr-anaglyph": dict(output_format="Red-Cyan Anaglyph"),
               "4k-dibr-vr": dict(output_format="VR"), "1080p-gui-defaults": GUI_KW, "4k-dibr-gui": GUI_KW,
               "4k-dibr-gui-hsbs": dict(GUI_KW, output_format="Half-SBS")}
HEADLINE = "4k-dav2b-dibr"
RENDER_KW = dict(output_format="Half-SBS", fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0,
                 feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)  # render_cli.py:24-33
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense f32-input MFMA (= vector rate) / dense bf16
VALU_SPEC_LANE_OPS = 78.6e12   # data sheet: 157.3 TFLOP/s float32 vector = 78.6 T lane-FMAs/s at 2.4 GHz (256 CUs x 4 SIMDs x 32 lanes / clock)
VALU_PEAK_LANE_OPS = 50.2e12   # MEASURED v_fma_f32 issue rate of the chip (tools/ubench_valu.hip, 8 waves / SIMD: 100.3 TFLOP/s = 50.2 T lane-FMAs/s;
                               # a wave64 VALU instruction issues every ~2.3 cycles per SIMD at the 1.77 GHz the chip sustains under that load --
                               # the data sheet's 157 TFLOP/s assumes 2.4 GHz).  Until round 3 this was 256 x 4 x 16 x 2.4e9 = 39.3 T, which the
                               # micro-benchmark itself exceeds.


def cpu_baseline(sh, sw, seconds_budget=20.0, max_frames=30):
    """The CPU oracle (kind "port", 1 core) on a bounded sample of the same frames; DIBR chain only."""
    from oracle import oracle as O
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.params import render_kwargs_to_params
    p = render_kwargs_to_params(sw, sh, output_height=sh, **RENDER_KW)
    ro = O.RenderOracle(p)
    ro.new_clip()
    n, t_used = 0, 0.0
    while n < 2 or (t_used < seconds_budget and n < max_frames):
        f, d = synth.synth_frame(n, sh, sw)
        t0 = time.perf_counter()
        ro.render(f, d, 0)
        t_used += time.perf_counter() - t0
        n += 1
        if t_used > seconds_budget:
            break
    return {"value": round(n / t_used, 4), "unit": "stereo-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames {sw}x{sh} Half-SBS through oracle/vd3d_oracle.c (DIBR chain only, precomputed depth; "
                      f"the oracle has no depth net), {t_used:.1f} s"}


def _cpu_worker(args):
    """One host core: an independent clip through the oracle (frames pre-generated outside the timed part)."""
    sh, sw, n_frames, wid = args
    from oracle import oracle as O
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.params import render_kwargs_to_params
    p = render_kwargs_to_params(sw, sh, output_height=sh, **RENDER_KW)
    ro = O.RenderOracle(p)
    ro.new_clip()
    clip = [synth.synth_frame(wid * 100 + i, sh, sw) for i in range(n_frames)]
    t0 = time.perf_counter()
    for f, d in clip:
        ro.render(f, d, 0)
    return time.perf_counter() - t0


def _host_mem_available_bytes():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) * 1024
    except Exception:
        pass
    return None


def _physical_cores():
    """Physical cores of the host (unique (package, core) pairs of /proc/cpuinfo restricted to this process's affinity mask); None if unreadable."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpu, phys = set(), None, None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k = k.strip()
            if k == "processor":
                cpu, phys = int(v), None
            elif k == "physical id":
                phys = int(v)
            elif k == "core id" and cpu in allowed:
                seen.add((phys, int(v)))
        return len(seen) or None
    except Exception:
        return None


def cpu_baseline_allcores(sh, sw, frames_per_core, max_cores=64, timeout_s=90.0):
    """The same oracle on every host thread at once (one independent clip per PROCESS, `bench.py --cpu-worker ...`): the CPU path's
    whole-socket throughput.  Plain subprocesses with a hard timeout -- nothing here can hang the benchmark.
    How many processes (round 6, measured on the MI355X box: 128 cores, 256 hardware threads, 4K frames): 64 processes -> 2.21 pairs/s (slowest 29 s per frame);
    128 (one per physical core) -> 1.59 pairs/s (slowest 81 s: the oracle's float32 planes stream through memory and the sockets' bandwidth is what saturates, not the
    cores); 256 (one per hardware thread) -> 103 of 256 workers had not finished their frame after 120 s.  The CPU path's best rate is therefore at 64 processes, and that
    is what `max_cores` defaults to: the baseline reports the CPU at its best, `cores` is what ran, `host_threads` what os.cpu_count() reports.  Half of the host's available
    memory must hold the processes (a 4K one peaks at ~1.03 GB, a 1080p one at ~0.3 GB: measured), else fewer run."""
    import subprocess
    host = os.cpu_count() or 1
    cores = min(_physical_cores() or host, max_cores or host)
    per_proc = 1.1e9 * (sh * sw) / (2160 * 3840) + 0.15e9
    avail = _host_mem_available_bytes()
    if avail:
        cores = max(1, min(cores, int(0.5 * avail / per_proc)))
    t0 = time.perf_counter()
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(sh), str(sw), str(frames_per_core), str(w)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for w in range(cores)]
    times = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=max(1.0, timeout_s - (time.perf_counter() - t0)))
            times.append(float(out.strip().splitlines()[-1]))
        except Exception:
            pr.kill()
    wall = time.perf_counter() - t0
    if len(times) < cores:
        return {"error": f"{cores - len(times)} of {cores} workers did not finish within {timeout_s:.0f} s"}
    tmax = max(times)
    return {"value": round(cores * frames_per_core / tmax, 3), "unit": "stereo-pairs/s", "cores": cores, "host_threads": host, "kind": "port",
            "sample": f"{frames_per_core} frames {sw}x{sh} per core on {cores} processes (independent clips, oracle/vd3d_oracle.c, DIBR chain "
                      f"only); slowest process {tmax:.1f} s, {wall:.1f} s wall incl. start-up"}


def cpu_depth_net(model_name, sh, sw):
    """The depth net of the headline on the HOST cores (torch CPU float32, all threads -- what the reference's own CPU path runs,
    core/render_depth.py:758-759 with device 'cpu'): one frame after one warm-up frame."""
    import torch
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import DepthPipe
    pipe = DepthPipe(model_name, device="cpu", dtype=torch.float32)
    f = torch.from_numpy(synth.synth_frame(0, sh, sw)[0])[None]
    pipe.infer_bgr_u8(f, raw=True)
    t0 = time.perf_counter()
    pipe.infer_bgr_u8(f, raw=True)
    dt = time.perf_counter() - t0
    return {"seconds_per_frame": round(dt, 3), "cores": torch.get_num_threads(), "kind": "torch-cpu float32 (same module graph)",
            "sample": f"1 frame {sw}x{sh} -> {model_name} on the host cores"}


class Env:
    """torch / distributed context shared by the workloads of one invocation"""

    def __init__(self, args, cpu_only=False):
        """cpu_only: gloo ranks on CPU tensors -- set only by the launcher / protocol test driver under tests/ (tests/bench_oracle_gloo.py), which
        re-uses this class, `self_launch` and the record helpers; nothing in this file can select it."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != max(args.gpus, 1):
            raise SystemExit(f"WORLD_SIZE={self.world} but --gpus {args.gpus}")
        self.cpu_only = bool(cpu_only)
        have_gpu = torch.cuda.is_available() and not self.cpu_only
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl" if have_gpu else "gloo", rank=self.rank, world_size=self.world)
        if not have_gpu and not self.cpu_only:
            if self.world > 1:
                # no GPU here: prove that the launcher started every rank and that they can talk, then stop cleanly (exit code 0 so
                # that "does `bench.py --gpus N` start N ranks" can be checked without hardware; there is still no CPU fallback)
                seen = [None] * self.world
                dist.all_gather_object(seen, (self.rank, os.getpid()))
                if self.rank == 0:
                    print(json.dumps({"error": "no GPU visible: bench.py needs MI355Xs (the HIP path has no CPU fallback)",
                                      "n_gpus": self.world, "ranks_started": len({r for r, _ in seen}), "rank_pids": [p for _, p in seen],
                                      "rendezvous": "gloo over 127.0.0.1 (the GPU run uses nccl = RCCL)"}), flush=True)
                dist.destroy_process_group()
                raise SystemExit(0)
            raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
        if have_gpu:
            if self.local_rank >= torch.cuda.device_count():
                raise SystemExit(f"rank {self.rank}: local rank {self.local_rank} but only {torch.cuda.device_count()} GPUs are visible")
            torch.cuda.set_device(self.local_rank)
        self.rank_pids = [os.getpid()]           # one process per rank: the record names them (config.rank_pids)
        if self.world > 1:
            seen = [None] * self.world
            dist.all_gather_object(seen, (self.rank, os.getpid(), self.local_rank))
            self.rank_pids = [p for _, p, _ in sorted(seen)]

    def fence(self):
        if not self.cpu_only:
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        if not self.cpu_only:
            self.torch.cuda.synchronize()


_CLIP_CACHE = {}


def _synth_cached(synth, t, sh, sw):
    """synthetic frames are pure functions of (index, size) and cost seconds of host time at 4K: the workloads of one invocation share them"""
    k = (t, sh, sw)
    if k not in _CLIP_CACHE:
        _CLIP_CACHE[k] = synth.synth_frame(t, sh, sw)
    return _CLIP_CACHE[k]


_P1_WAIT_MIN = [None]


def _p1_wait(env, sharders):
    """P1-chain wait per step (sharded.ChunkSharder.p1_wait_ms), the maximum over slot sets and ranks (rank 0 never waits inside a step)."""
    if not sharders or env.world == 1:
        return None
    v = max([s.p1_wait_ms() or 0.0 for s in sharders])
    t = env.torch.tensor([v, -v], dtype=env.torch.float64, device="cpu" if env.cpu_only else "cuda")
    env.dist.all_reduce(t, op=env.dist.ReduceOp.MAX)
    _P1_WAIT_MIN[0] = round(-float(t[1].item()), 4)     # the smallest per-rank figure (rank 0 never waits inside a step)
    return round(float(t[0].item()), 4)


def run_workload(env: Env, args, workload: str, steps: int, warmup: int, depth_dtype: str = "f32", profile: bool = True,
                 isolated_pass: bool = True, host_io=None):
    """Time `steps` steps of one workload after `warmup` untimed steps.  Returns a dict with the timing, the HIP-event stage
    averages taken in a profiled pass behind the timed region (round 6), and the parameters needed to price them."""
    torch, dist = env.torch, env.dist
    rank, world, local_rank = env.rank, env.world, env.local_rank
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.params import render_kwargs_to_params
    from visiondepth3d_amd.render_3d import Renderer

    sh, sw, model_name, desc = WORKLOADS[workload]
    if host_io is None:
        host_io = "nv12" if args.host_io_nv12 else bool(args.host_io)
    p = render_kwargs_to_params(sw, sh, output_height=sh, dof_dense_conv=not workload.endswith("-sepdof"),
                                **dict(RENDER_KW, **WORKLOAD_KW.get(workload, {})))
    overlap = (not args.no_overlap) and model_name is not None
    r = Renderer(local_rank, private_stream=overlap, auto_order=False)   # bench orders its streams by hand; DIBR chain on its own stream when a depth net shares the GPU
    rh = Renderer(local_rank) if overlap else r        # depth hand-off stays on the depth net's (torch) stream
    dibr_stream = r.stream if overlap else None
    r.new_clip()
    B = args.batch

    # ONE synthetic clip, cut into contiguous chunks across the ranks (sharded.py): step k holds the clip's frames
    # [k * world * B, (k + 1) * world * B) and rank g owns [g * B, (g + 1) * B) of them, so this rank's local frame L is the clip's frame
    # (L // B) * world * B + g * B + L % B.  Only the own frames are resident in this rank's HBM (world 1: frames 0 .. clip - 1).
    gidx = [(L // args.batch) * world * args.batch + rank * args.batch + L % args.batch for L in range(args.clip)]
    frames_np, depths_np = zip(*[_synth_cached(synth, t, sh, sw) for t in gidx])
    frames = torch.stack([torch.from_numpy(f) for f in frames_np]).cuda()          # [C,h,w,3] u8
    depths = torch.stack([torch.from_numpy(d) for d in depths_np]).cuda()          # [C,h,w] f32
    outs = torch.empty((B, p.out_h, p.out_w, 3), dtype=torch.uint8, device="cuda")
    ring = h_clip = None
    nv12 = host_io == "nv12"    # NV12 wire format both ways (video_io.PIPE_PIX_FMT = "nv12"): 1.5 bytes per pixel cross PCIe instead of 3
    fb_bgr = outs_bgr = None
    if host_io:   # the clip lives in pinned host memory, results return to pinned host memory
        from visiondepth3d_amd.frame_io import PinnedRing
        if nv12:
            conv = Renderer(local_rank)
            h_clip = torch.stack([conv.bgr_to_nv12(frames[j]) for j in range(frames.shape[0])]).cpu().pin_memory()   # [C, h*3/2, w]
            conv.close()
            ring = PinnedRing(B, (sh * 3 // 2, sw), (p.out_h * 3 // 2, p.out_w), torch.device("cuda", local_rank), depth=int(args.ring_depth))
            fb_bgr = torch.empty((B, sh, sw, 3), dtype=torch.uint8, device="cuda")
            outs_bgr = torch.empty((B, p.out_h, p.out_w, 3), dtype=torch.uint8, device="cuda")
        else:
            h_clip = frames.cpu().pin_memory()
            ring = PinnedRing(B, (sh, sw, 3), (p.out_h, p.out_w, 3), torch.device("cuda", local_rank), depth=int(args.ring_depth))

    # host-io: measured slower with the overlapped passes (the copy engines then compete with two compute streams)
    pix_ov = (not host_io) if args.pixel_overlap is None else bool(args.pixel_overlap)
    if nv12:
        pix_ov = False    # the colour conversions run on the renderer's one stream, in front of and behind the step
    shr2 = None
    if world > 1 or args.sharded or pix_ov or not args.per_frame:   # the step protocol (batched select chain) is the default DIBR path at any world size
        from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend
        be = HipChunkBackend(r, p)
        shr2 = [ChunkSharder(be, rank, world, B)]
        if pix_ov:
            shr2.append(ChunkSharder(be, rank, world, B, slot_base=B, twin_of=shr2[0]))
            r.set_pixel_overlap(1 if host_io else max(1, int(args.pix_streams)))   # frames' pixel passes round-robin over this many streams

    pipe = None
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f32x3": torch.float32, "f32h2": torch.float32}[depth_dtype]
    if model_name:
        from visiondepth3d_amd.depth import DepthPipe
        # fused front end + fused backbone / neck glue; library selection: committed GEMM table + MIOpen find mode (runs during the warm-up)
        # (find mode only where the number is a parity-mode one: the bf16 sub-record keeps MIOpen's immediate mode and its shorter start)
        pipe = DepthPipe(model_name, device="cuda", dtype=tdt, renderer=rh, miopen_find=(not args.no_miopen_find) and depth_dtype in ("f32", "f32x3", "f32h2"),
                         gemm={"f32x3": "bf16x3", "f32h2": "fp16x2"}.get(depth_dtype, "f32"))

    NBUF = 2  # double-buffered hand-off planes so batch i+1's depth inference overlaps batch i's DIBR chain
    dbuf = [torch.empty((B, sh, sw), dtype=torch.uint8, device="cuda") for _ in range(NBUF)]
    done = [torch.cuda.Event() for _ in range(NBUF)]
    net_ev = []   # (start, end) torch events around the depth net + hand-off of the timed steps
    step_no = [0]

    def step(i, timed=False):
        shr = shr2[i % len(shr2)] if shr2 else None    # pixel overlap: alternate slot sets so the next chain never waits for these pixels
        idx = [(i * B + j) % args.clip for j in range(B)]
        contiguous = idx == list(range(idx[0], idx[0] + B))
        fb = frames[idx[0]:idx[0] + B] if contiguous else frames[idx]
        k = i % NBUF
        outs_k = outs
        if ring is not None:
            kr = i % ring.n
            ring.upload(kr, h_clip[idx[0]:idx[0] + B] if contiguous else h_clip[idx])
            fb, outs_k = ring.d_in[kr], ring.d_out[kr]
            if nv12:   # NV12 -> BGR on the device (vd3d_nv12_to_bgr), the step renders into a BGR buffer, BGR -> NV12 below
                for j in range(B):
                    r.nv12_to_bgr(ring.d_in[kr][j], out=fb_bgr[j])
                fb, outs_k = fb_bgr, outs_bgr
        if overlap:
            torch.cuda.current_stream().wait_event(done[k])  # hand-off buffer k is free again (no-op until first recorded)
        dloc = None
        if pipe is not None:
            if timed and profile:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            pred = pipe.infer_bgr_u8(fb, raw=True)              # [B,518,924] f32 on device
            dloc = rh.depth_handoff(pred, sh, sw, out=dbuf[k])  # the reference's 8-bit depth hand-off (a24), fused, no disk hop
            if timed and profile:
                e1.record()
                net_ev.append((e0, e1))
        if shr is not None and dloc is None:
            dloc = depths[idx[0]:idx[0] + B] if contiguous else depths[idx]   # the precomputed f32 planes
        if overlap:  # hand the batch to the DIBR stream; this (torch) stream goes on to the next batch's depth inference
            ev = torch.cuda.Event()
            ev.record()
            dibr_stream.wait_event(ev)
        if ring is not None:
            if overlap:
                dibr_stream.wait_event(ring.ev_in[kr])
            ring.reserve_output(kr, compute_stream=r.pixel_stream if pix_ov else dibr_stream)
        if shr is None:
            for j in range(B):
                r.render_frame(fb[j], dloc[j] if dloc is not None else depths[idx[j]], p, out=outs_k[j])
        else:   # chunked sharding (visiondepth3d_amd/sharded.py): the point-to-point plane hand-off and the two record all-gathers
                # are enqueued on the DIBR stream, where the planes / records are produced and consumed
            on_dibr = (lambda: torch.cuda.stream(dibr_stream)) if overlap else contextlib.nullcontext
            with on_dibr():
                shr.p1(fb, dloc, first_step=(step_no[0] == 0), more_steps=True)
                q_all = shr.gather(shr.q_local)
            shr.r1(q_all)
            shr.p3()
            with on_dibr():
                m_all = shr.gather(shr.m_local)
            shr.r2(m_all)
            shr.pixels(outs_k)
        step_no[0] += 1
        if overlap:
            done[k].record(dibr_stream)
        if ring is not None:   # D2H behind the stream that produced the muxed frames
            if nv12:
                for j in range(B):
                    r.bgr_to_nv12(outs_bgr[j], out=ring.d_out[kr][j])
            ring.download(kr, compute_stream=r.pixel_stream if pix_ov else dibr_stream)

    for i in range(warmup):
        step(i)
    env.fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    env.fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Per-stage HIP events: a PROFILED PASS of a few more steps behind the timed region (same streams, same overlap), not inside it.  Round 6: with the stage timers
    # on, every stage of every frame is bracketed by two event records, and the DIBR-only workloads measured 5 - 8 % under a run without them (4k-dibr-dof3 1 199
    # against 1 291 pairs/s) -- the product path has no such hooks, so `value` is taken without them; on the headline the difference is inside the noise.
    stage_ms = {}
    if profile:
        r.set_profiling(True)
        for i in range(max(2, min(steps, 6))):
            step(warmup + steps + i, timed=True)
        env.fence()
        for name in ("frame", "ingest", "select_eye", "select_dc", "shape", "select_s1", "shift", "w1", "e2w", "warp", "finish",
                     "p1_own", "p3_own", "replay"):
            v = r.stage_ms(name)
            if v >= 0:
                stage_ms[name] = round(v, 5)
        r.set_profiling(False)
    net_ms = None
    if net_ev:
        net_ms = sum(a.elapsed_time(b) for a, b in net_ev) / len(net_ev)

    # W1 / E1 alone on the GPU: a short sequential DIBR-only pass AFTER the timed region (same frames, same kernels)
    iso_ms = {}
    if rank == 0 and profile and isolated_pass and world == 1:
        r.set_profiling(True)   # clears the accumulators of the timed region (already read above)
        for j in range(min(B, 8, len(depths))):
            r.render_frame(frames[j], depths[j], p, out=outs[j])
        r.sync()
        iso_ms = {"w1": round(r.stage_ms("w1"), 5), "e2w": round(r.stage_ms("e2w"), 5), "finish": round(r.stage_ms("finish"), 5),
                  "frame": round(r.stage_ms("frame"), 5)}
        r.set_profiling(False)

    flops = pipe.flops_per_frame(sh, sw) if (pipe is not None and rank == 0) else None
    feather_on = bool(p.shift.enable_feathering) and float(p.shift.feather_strength) > 0.0
    # W1's algorithmic bytes per stereo pair: the eye-resolution RGB planes (3 x float32) + one warp-resolution float32 plane (the
    # shift / depth the samples are steered by) + two uint8 eyes out: 3 N + 4 N + 6 N = 13 N when the eyes are half the warp size each way
    w1_alg = 12 * p.eye_h * p.eye_w + 4 * p.warp_h * p.warp_w + 6 * p.warp_h * p.warp_w
    res = dict(workload=workload, desc=desc, sh=sh, sw=sw, model=model_name, B=B, steps=steps, warmup=warmup, dt=dt,
               w1_alg_bytes=w1_alg, e1_alg_bytes=6 * p.warp_h * p.warp_w + 4 * p.eye_h * p.eye_w + 3 * p.out_h * p.out_w, feather_on=feather_on, eye=(p.eye_h, p.eye_w), warp=(p.warp_h, p.warp_w), out=(p.out_h, p.out_w),
               frames_total=world * steps * B, stage_ms=stage_ms, iso_ms=iso_ms, net_ms=net_ms, flops_per_frame=flops,
               N=p.warp_h * p.warp_w, pix_ov=bool(pix_ov), pix_streams=(1 if host_io else max(1, int(args.pix_streams))) if pix_ov else 0, ring_depth=(int(args.ring_depth) if host_io else None),
               depth_dtype=depth_dtype if model_name else None,
               lib_sel=({"hipblaslt_solution_table": bool(pipe.tuned_gemm), "miopen_find_mode": bool(pipe.miopen_find)} if pipe is not None else None),
               host_io=bool(host_io), clip=args.clip, clip_frames_global=world * args.clip,
               p1_wait_ms=_p1_wait(env, shr2),
               shard_bytes=(shr2[0].bytes_per_step() if (shr2 and hasattr(shr2[0], "bytes_per_step")) else None))
    del pipe
    r.close()
    if rh is not r:
        rh.close()
    import gc
    gc.collect()   # the fused-backbone rewrites are closures on the model's own layers (reference cycles): collect the whole pipe NOW, so that neither its
                   # device memory nor its tens of thousands of tracked Python objects ride along through the later workloads' launch loops
    torch.cuda.empty_cache()
    return res


def run_upscale_chain(env, args, steps=4, warmup=2, B=8):
    """BASELINE configs[4] on one GPU: 1080p frames -> DA-V2-Small (float32) -> DIBR Half-SBS 1920x1080 -> run_esrgan(input_res_pct=50,
    target_size=(3840, 2160)) with the reference's default model (RealESR_Gx4_fp16, fp16 like its ONNX export).  Two streams (round 4): the
    up-scale net of batch i runs behind depth + DIBR of batch i + 1 (each side with its own renderer context, two stereo-frame buffers,
    events both ways); `--chain-serial`: everything on one stream.  A sub-record, never the headline."""
    torch = env.torch
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import DepthPipe
    from visiondepth3d_amd.params import render_kwargs_to_params
    from visiondepth3d_amd.render_3d import Renderer
    from visiondepth3d_amd.upscale import Upscaler
    sh, sw = 1080, 1920
    p = render_kwargs_to_params(sw, sh, output_height=sh, **RENDER_KW)
    serial = bool(getattr(args, "chain_serial", False))
    r = Renderer(env.local_rank)
    r.new_clip()
    r_up = r if serial else Renderer(env.local_rank)      # the up-scale side's own context: the two streams never share context scratch
    pipe = DepthPipe("depth-anything-v2-small", device="cuda", dtype=torch.float32, renderer=r, miopen_find=False)   # the chain is the up-scale net's: no find pass
    up = Upscaler(r_up, "RealESR_Gx4_fp16")
    frames_np, _ = synth.synth_clip(B, sh, sw, start=0)
    frames = torch.stack([torch.from_numpy(f) for f in frames_np]).cuda()
    dbuf = torch.empty((B, sh, sw), dtype=torch.uint8, device="cuda")
    NB = 1 if serial else 2
    outs_set = [torch.empty((B, p.out_h, p.out_w, 3), dtype=torch.uint8, device="cuda") for _ in range(NB)]
    outs = outs_set[0]
    s_dibr = torch.cuda.current_stream() if serial else torch.cuda.Stream()
    s_up = s_dibr if serial else torch.cuda.Stream()
    if not serial:   # everything set up so far (weights, clip, buffers) was enqueued on the current stream
        s_dibr.wait_stream(torch.cuda.current_stream())
        s_up.wait_stream(torch.cuda.current_stream())
    dibr_done = [torch.cuda.Event() for _ in range(NB)]
    up_done = [None] * NB
    ev = []
    last = [None]

    from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend
    shr = ChunkSharder(HipChunkBackend(r, p), 0, 1, B)       # the DIBR frames of a step go through the batched step path (no pixel overlap: serial chain)
    nstep = [0]

    def step(timed):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        k = nstep[0] % NB
        o = outs_set[k]
        with torch.cuda.stream(s_dibr):
            if up_done[k] is not None:
                s_dibr.wait_event(up_done[k])          # the up-scale net of two steps ago has read this buffer
            e[0].record(s_dibr)
            pred = pipe.infer_bgr_u8(frames, raw=True)
            r.depth_handoff(pred, sh, sw, out=dbuf)
            e[1].record(s_dibr)
            shr.render_step(frames, dbuf, outs=o, first_step=(nstep[0] == 0))
            e[2].record(s_dibr)
            dibr_done[k].record(s_dibr)
        nstep[0] += 1
        with torch.cuda.stream(s_up):
            s_up.wait_event(dibr_done[k])
            e[3].record(s_up)
            for j in range(B):
                last[0] = up.run_esrgan(o[j], input_res_pct=50, target_size=(3840, 2160))
            e[4].record(s_up)
            up_done[k] = torch.cuda.Event()
            up_done[k].record(s_up)
        if timed:
            ev.append(e)

    for _ in range(warmup):
        step(False)
    env.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    env.fence()
    dt = time.perf_counter() - t0
    ms = [sum(e[i].elapsed_time(e[i + 1]) for e in ev) / len(ev) / B for i in (0, 1, 3)]   # per-stream HIP events; with two streams the stages overlap
    # the network alone, for its MFMA figure
    flops = [0.0]

    def hook(mod, inp, out):
        flops[0] += 2.0 * out.numel() * (mod.in_channels // mod.groups) * mod.kernel_size[0] * mod.kernel_size[1]

    hs = [m.register_forward_hook(hook) for m in up.net.modules() if isinstance(m, torch.nn.Conv2d)]
    x = r.esr_preprocess(r.resize_area_u8(outs[0], 540, 960), dtype=up.dtype)
    with torch.no_grad():
        up.net(x)          # flop count only (the module graph has the same convolutions as the product path)
    for h in hs:
        h.remove()

    def timed(fn, n=5):
        with torch.no_grad():
            fn(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        with torch.no_grad():
            for _ in range(n):
                fn(x)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    net_ms = timed(up._forward)        # the PRODUCT path: every layer hand-written (head, 32 body layers + tail convolution on the MFMA kernel, pixel-shuffle + add)
    lib_ms = timed(up.net)             # the same network through the library convolutions only (context, not the product)
    out_shape = list(last[0].shape)
    del pipe, up
    if r_up is not r:
        r_up.close()
    r.close()
    torch.cuda.empty_cache()
    return {"workload": "1080p-dav2s-dibr-esrgan4k",
            "description": "BASELINE configs[4] on one GPU: 1080p, DA-V2-Small (float32) + DIBR Half-SBS + Real-ESRGAN x4 (RealESR_Gx4, fp16 like "
                           "the reference's ONNX export) through run_esrgan(input_res_pct=50, target_size=(3840, 2160)); " +
                           ("serial chain, one stream" if serial else "two streams: the up-scale net of batch i behind depth + DIBR of batch i + 1") + " (DIBR through the batched step path)",
            "streams": 1 if serial else 2,
            "value": round(steps * B / dt, 3), "unit": "stereo-pairs/s", "steps": steps, "warmup": warmup, "frames_timed": steps * B,
            "ms_per_step": round(dt / steps * 1e3, 3), "dtype": "f32 depth net + f32 DIBR + fp16 up-scale net (the reference's precisions)",
            "output": out_shape, "stage_ms_per_frame": {"depth_net+handoff": round(ms[0], 3), "dibr": round(ms[1], 3), "run_esrgan": round(ms[2], 3)},
            "roofline_upscale_net": {"bound": "mfma", "kernel": "SRVGGNetCompact 64x32 on a 960x540 frame, Upscaler._forward = the product path: 32 body "
                                     "layers and the 64->48 tail convolution on the hand-written k_conv3x3_c64 (MFMA, fp16), head 3->64 (k_conv3x3_head) and pixel-shuffle + nearest add (k_esr_tail) in HIP: no library call",
                                     "achieved": round(flops[0] / (net_ms * 1e-3) / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                     "frac": round(flops[0] / (net_ms * 1e-3) / 1e12 / 2500.0, 4), "flops_per_frame": flops[0],
                                     "avg_forward_ms": round(net_ms, 3), "library_only_forward_ms": round(lib_ms, 3)}}


def self_launch(n, script=None):
    """`python bench.py --gpus N` without a launcher: re-execute this command line under torch.distributed.run, one rank per GPU of this
    node (rendezvous on 127.0.0.1, a free port; HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only does dmabuf IPC).  Rank 0 prints the
    JSON line on the inherited stdout; the exit code is the launcher's.  `script`: the file to re-execute (default: this one; the protocol
    test driver under tests/ launches itself through the same code)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script or __file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def copy_yardstick(env):
    """measured streaming-copy rate: device-to-device copy of 1 GiB through the library's own copy kernel"""
    torch = env.torch
    from visiondepth3d_amd.render_3d import Renderer
    r = Renderer(env.local_rank)
    nbytes = 1 << 30
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    for _ in range(3):
        r._L.vd3d_stream_copy(r._ctx, a.data_ptr(), b.data_ptr(), nbytes)
    r.set_profiling(True)
    for _ in range(10):
        r._L.vd3d_stream_copy(r._ctx, a.data_ptr(), b.data_ptr(), nbytes)
    ms = r.stage_ms("stream_copy")
    r.set_profiling(False)
    r.close()
    return 2 * nbytes / (ms * 1e-3) / 1e9


def _git_commit():
    """Short hash of the tree the line was measured on (None on a snapshot without .git, e.g. the GPU box)."""
    import subprocess
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip() or None
    except Exception:
        return None


def _pmc(workload):
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        return pm.get(workload)
    except Exception:
        return None


def rooflines(res, copy_gbs=None, pmc_workload=None):
    """roofline objects of one measured workload: W1 (north_star's kernel), E1, the whole DIBR frame, the depth net."""
    out = {}
    N, st, iso = res["N"], res["stage_ms"], res["iso_ms"]
    pm = _pmc(pmc_workload or res["workload"]) or {}
    # Which duration prices a kernel.  Since round 4 the DIBR step runs on three streams (select chain + two pixel streams): inside the timed
    # region a launch of W1 / E1 shares the CUs with other kernels, so its HIP-event duration grows with the concurrency while the frame rate
    # RISES -- it is no longer the kernel's cost.  `frac` / `achieved` / `avg_launch_ms` therefore come from the SEQUENTIAL pass over the same
    # frames that follows the timed region (HIP events, one kernel on the GPU at a time: what `rocprofv3 --kernel-trace --stats` of
    # `bench.py --workload 4k-dibr --no-pixel-overlap` reports, profiles/rNN_4k_dibr_kernel_stats.md); the in-step durations stay next to them.
    def pick(key):
        seq, instep = iso.get(key, -1) or -1, st.get(key, -1)
        return (seq, instep, "sequential pass after the timed region") if seq > 0 else (instep, instep, "profiled pass behind the timed region (no sequential pass in this run)")
    w1_ms, w1_instep, w1_src = pick("w1")
    if w1_ms > 0:
        alg = res.get("w1_alg_bytes", 13 * N)  # W1 = read RGB 3N + read depth 4N + write two u8 eyes 6N per stereo pair (13 N with half-size eyes)
        ach = alg / (w1_ms * 1e-3) / 1e9
        e2w_ms = (iso.get("e2w") if w1_src.startswith("sequential") else st.get("e2w")) or -1
        w1 = pm.get("k_warp_fused", pm if "corrected_bytes_per_launch" in pm else {})
        lane = w1.get("valu_lane_instr_per_launch")
        w1_kernel = ("W1 = k_e2w (warped-depth gradient mask of both eyes) + k_warp_fused (window sums + warp + blend): the same work as round 3's "
                     "single launch, two launches since round 4") if res.get("feather_on", True) else \
                    ("W1 = k_warp_fused<FEATHER = false, SHIFT = true> alone: feather_strength <= 0 makes feather_shift_edges an exact no-op (round 5), no mask "
                     "kernel, no window sums, no blend -- nested-bilinear warp of both eyes + truncation; since round 6 the launch also computes the shift "
                     "values of its own tile (k_shift folded in: it reads the shaped depth, 4 N, where it used to read the shift plane), so this duration "
                     "compares with k_shift + W1 of earlier records (60 + 66 us at 4K)")
        rf = {"bound": "valu" if lane else "hbm", "kernel": w1_kernel,
              "k_e2w_avg_launch_ms": (e2w_ms if e2w_ms > 0 else None), "k_warp_fused_avg_launch_ms": (round(w1_ms - e2w_ms, 5) if e2w_ms > 0 else None),
              "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
              "traffic": w1.get("corrected_bytes_per_launch"), "traffic_source": w1.get("source"),
              # FETCH_SIZE at face value + WRITE_SIZE: the guide's x2 read correction is calibrated on 16-byte-per-lane streaming reads; the no-feather W1's
              # 4-byte-per-lane reads are counted at face value (raw FETCH 57.75 MB = its 3 N + 4 N of reads), so for THAT kernel this is the true figure
              "traffic_uncorrected": (int(round((w1["fetch_size_kb_raw"] + w1["write_size_kb"]) * 1000)) if w1.get("fetch_size_kb_raw") and w1.get("write_size_kb") else None),
              "traffic_taken_at_commit": (_pmc("commit") or None),
              "algorithmic_bytes_per_launch": alg, "avg_launch_ms": w1_ms, "avg_launch_measured": w1_src,
              "rocprof_avg_launch_ms": (round(w1["rocprof_avg_launch_us"] / 1e3, 5) if w1.get("rocprof_avg_launch_us") else None),
              "in_step_avg_launch_ms": w1_instep if w1_instep > 0 else None,   # (round 6: from the profiled pass behind the timed region)
              "in_step_frac": round(alg / (w1_instep * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if w1_instep > 0 else None,
              "measured_copy_GBs": round(copy_gbs, 1) if copy_gbs else None,
              "note": "frac prices SURVEY 8(d)'s 13 N algorithmic bytes against the 8 TB/s HBM spec (north_star's yardstick); the kernel is "
                      "neither HBM- nor VALU-bound (the reference's nested-bilinear arithmetic is kept bit-exact; round 6's ablations, profiles/r06_w1_phases.md: "
                      "what a workgroup waits for are dependent LDS / scalar-branch round trips at 2.5 resident workgroups per CU): `valu` prices the "
                      "PMC-counted VALU lane-instructions of one launch against the chip's MEASURED v_fma_f32 issue rate (50.2 T lane-ops/s) and the data sheet's. "
                      "ONE source prices the kernel: avg_launch_ms / achieved / frac = HIP events of the sequential pass of THIS run; rocprof_avg_launch_ms is the "
                      "committed rocprofv3 --kernel-trace figure of the same command (profiles/pmc_latest.json, taken at traffic_taken_at_commit), printed for "
                      "cross-checking only. avg_launch_ms / achieved / frac: HIP events of the sequential pass over the same frames that follows the timed region (one kernel on "
                      "the GPU at a time: the kernel's cost, what rocprofv3 --kernel-trace of the --no-pixel-overlap run reports); in_step_*: HIP events "
                      "of a profiled pass of the same steps behind the timed region (round 6: not inside it), where the launch shares the CUs with the chain and the other pixel stream (grows with the concurrency "
                      "while the frame rate rises). traffic is that of BOTH launches: the E2 plane's round trip (8 N written, 8 N + halo read) is the "
                      "price of taking the dependent depth gathers out of W1's tile"}
        if lane:
            t = w1_ms * 1e-3
            rf["valu"] = {"lane_instr_per_pixel": round(lane / N, 1), "measured_v_fma_rate_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                          "spec_lane_ops_per_s": VALU_SPEC_LANE_OPS, "frac_of_measured_rate": round(lane / t / VALU_PEAK_LANE_OPS, 4),
                          "frac_of_spec_rate": round(lane / t / VALU_SPEC_LANE_OPS, 4), "source": w1.get("source")}
            rf["valu_frac_of_spec"] = rf["valu"]["frac_of_spec_rate"]   # beside `frac` (HBM yardstick) in the same object: the bound that actually binds
        if alg and rf.get("traffic"):
            rf["traffic_over_algorithmic"] = round(rf["traffic"] / alg, 3)
        out["roofline"] = rf
    fin_ms, fin_instep, fin_src = pick("finish")
    if fin_ms > 0:  # E1: 6N eyes in + N eye-res depth + 3N Half-SBS out (10 N at 4K Half-SBS; per geometry: two u8 eyes + the eye-res float32 depth + the muxed frame)
        alg = res.get("e1_alg_bytes", 10 * N)
        ach = alg / (fin_ms * 1e-3) / 1e9
        e1 = pm.get("k_finish_fused", {})
        lane = e1.get("valu_lane_instr_per_launch")
        rf = {"bound": "valu" if lane else "hbm", "kernel": "k_finish_fused (E1: DOF + grade + sharpen + fit + mux, both eyes)",
              "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
              "traffic": e1.get("corrected_bytes_per_launch"), "algorithmic_bytes_per_launch": alg, "avg_launch_ms": fin_ms,
              "avg_launch_measured": fin_src, "in_step_avg_launch_ms": fin_instep if fin_instep > 0 else None,
              "rocprof_avg_launch_ms": (round(e1["rocprof_avg_launch_us"] / 1e3, 5) if e1.get("rocprof_avg_launch_us") else None),
              "in_step_frac": round(alg / (fin_instep * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if fin_instep > 0 else None}
        if lane:   # E1 against BOTH bounds (HBM above, fp32 ALU here)
            t = fin_ms * 1e-3
            rf["valu"] = {"lane_instr_per_pixel": round(lane / N, 1), "measured_v_fma_rate_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                          "spec_lane_ops_per_s": VALU_SPEC_LANE_OPS, "frac_of_measured_rate": round(lane / t / VALU_PEAK_LANE_OPS, 4),
                          "frac_of_spec_rate": round(lane / t / VALU_SPEC_LANE_OPS, 4), "source": e1.get("source")}
        out["roofline_e1"] = rf
    fr_ms, note = st.get("frame", -1), "sequential frame: K1-K6, k_shift, W1, E1"
    if fr_ms <= 0 and all(st.get(k, -1) > 0 for k in ("p1_own", "p3_own", "warp", "finish")):
        fr_ms = (st["p1_own"] + st["p3_own"] + max(st.get("replay", 0.0), 0.0)) / res["B"] + st["warp"] + st["finish"]
        note = ("sum of the frame's stage durations: (P1 + P3 + replay) / B -- the select chain runs batched over the B frames of a step since round 4 -- "
                "+ k_shift + k_e2w + W1 + E1; chain and pixel kernels share the GPU on separate streams, so the stage durations overlap")
    if fr_ms > 0:  # whole DIBR chain = RGB 3N + depth 4N twice + two u8 eyes 6N = 17 N per stereo pair
        ch = 17 * N / (fr_ms * 1e-3) / 1e9
        out["roofline_chain"] = {"bound": "latency", "kernel": "whole DIBR frame (" + note + ")", "achieved": round(ch, 2),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ch / HBM_PEAK_GBS, 5),
                                 "algorithmic_bytes_per_frame": 17 * N, "avg_frame_ms": round(fr_ms, 5),
                                 "isolated_avg_frame_ms": iso.get("frame")}
    if res.get("net_ms") and res.get("flops_per_frame"):
        tf = res["flops_per_frame"] * res["B"] / (res["net_ms"] * 1e-3) / 1e12
        pk = MFMA_PEAK_TFLOPS[{"f32x3": "f32", "f32h2": "f32"}.get(res["depth_dtype"], res["depth_dtype"])]   # f32x3: float32-equivalent flops against the float32 MFMA peak (may exceed 1)
        out["roofline_depthnet"] = {"bound": "mfma", "kernel": f"{res['model']} forward + hand-off ({res['depth_dtype']}; hipBLASLt / AOTriton / "
                                    "MIOpen through PyTorch-ROCm, glue fused in HIP)", "achieved": round(tf, 2), "peak": pk,
                                    "unit": "TFLOP/s", "frac": round(tf / pk, 4), "flops_per_frame": res["flops_per_frame"],
                                    "avg_batch_ms": round(res["net_ms"], 3), "frames_per_batch": res["B"],
                                    "note": "torch-event time of depth inference + 8-bit hand-off per batch in the profiled pass behind the timed region, "
                                            "while the DIBR streams of the previous batch share the GPU"}
    return out


def sub_record(res, extra=None):
    d = {"workload": res["workload"], "description": res["desc"], "value": round(res["frames_total"] / res["dt"], 3),
         "unit": "stereo-pairs/s", "steps": res["steps"], "warmup": res["warmup"], "frames_timed": res["frames_total"],
         "ms_per_step": round(res["dt"] / res["steps"] * 1e3, 4),
         "dtype": {None: "f32", "f32": "f32", "f32x3": "f32 (bf16x3 split MFMA, f32 accumulate)", "f32h2": "f32 operands as 2 x fp16 (22 bits), MFMA, f32 accumulate"}.get(
             res["depth_dtype"], "f32 DIBR + bf16 depth net (REDUCED precision vs the reference's float32)")}
    if extra:
        d.update(extra)
    return d

COMPACT_LINE_LIMIT = 6000   # bytes; the driver reads the bench line out of an 8 KB tail of stdout (round 5's 22 KB line came back `parsed: null`)
_ROOF_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_uncorrected", "algorithmic_bytes_per_launch", "algorithmic_bytes_per_frame", "avg_launch_ms",
              "rocprof_avg_launch_ms", "in_step_avg_launch_ms", "k_e2w_avg_launch_ms", "k_warp_fused_avg_launch_ms", "k_shift_avg_launch_ms",
              "valu_frac_of_spec", "traffic_over_algorithmic", "measured_copy_GBs", "measured_on", "avg_frame_ms", "avg_batch_ms", "frames_per_batch",
              "flops_per_frame")
_ROOF_KERNEL = {"roofline": "W1", "roofline_e1": "E1 k_finish_fused", "roofline_chain": "whole DIBR frame", "roofline_depthnet": "depth net"}


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_record(res, full_path=None):
    """The ONE line the driver parses (bench.py's last stdout line), kept under COMPACT_LINE_LIMIT bytes: the contract's headline fields,
    `config`, the roofline objects reduced to their numbers, `cpu_baseline`, and sub-records as {name: {value, ms_per_step, dtype}}.
    Notes, per-sub-record rooflines and stage tables stay in the FULL record (printed on an earlier stdout line as
    {"bench_full_record": ...} and written to `full_path`)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: res[k] for k in keep if k in res}
    out["data"] = _short(out.get("data", "synthetic"), 120)
    cfg = res.get("config", {})
    ck = ("workload", "frame", "format", "frames_per_step", "depth_model", "depth_net_dtype", "pixel_overlap", "pixel_streams", "rccl_ranks",
          "rank_pids", "p1_chain_wait_ms_per_step_min_max", "comm_per_step_per_rank", "clip_frames_all_ranks", "aten_threads")
    out["config"] = {k: cfg[k] for k in ck if cfg.get(k) is not None}
    out["config"]["workload"] = cfg.get("workload")
    out["config"]["params"] = "render_cli.py defaults + dof_strength 2.0, dense DOF order (parity mode)"
    for rk, short in _ROOF_KERNEL.items():
        rf = res.get(rk)
        if not rf:
            continue
        o = {k: rf[k] for k in _ROOF_KEEP if rf.get(k) is not None}
        if rk in ("roofline", "roofline_e1") and "traffic" not in o:
            o["traffic"] = None   # the contract names the key
        o["kernel"] = _short(rf.get("kernel_short") or short, 80)
        if isinstance(rf.get("valu"), dict):
            o["valu_lane_instr_per_pixel"] = rf["valu"].get("lane_instr_per_pixel")
            o["valu_frac_of_measured_rate"] = rf["valu"].get("frac_of_measured_rate")
            o.setdefault("valu_frac_of_spec", rf["valu"].get("frac_of_spec_rate"))
        out[rk] = o
    cb = res.get("cpu_baseline")
    if cb:
        o = {k: cb[k] for k in ("value", "unit", "cores", "host_threads", "kind") if k in cb}
        o["sample"] = _short(cb.get("sample", ""), 200)
        if isinstance(cb.get("single_core"), dict):
            o["single_core_value"] = cb["single_core"].get("value")
        if isinstance(cb.get("gpu_same_work"), dict):
            o["gpu_same_work"] = {k: cb["gpu_same_work"].get(k) for k in ("workload", "value")}
        if "all_cores" in cb:
            o["all_cores_error"] = _short(cb["all_cores"].get("error", ""), 120)
        out["cpu_baseline"] = o
    c1 = res.get("cpu_baseline_1080p")
    if c1:
        out["cpu_baseline_1080p"] = {"value": c1.get("value"), "cores": c1.get("cores"),
                                     "gpu_same_work_value": (c1.get("gpu_same_work") or {}).get("value")}
    srs = res.get("sub_records")
    if srs:
        out["sub_records"] = {n: ({"value": r.get("value"), "ms_per_step": r.get("ms_per_step"), "dtype": _short(r.get("dtype", "f32"), 48)}
                                  if "error" not in r else {"error": _short(r["error"], 80)}) for n, r in srs.items()}
        # W1 where north_star's HBM question is meaningful (no feathering): the two numbers, nothing else
        for n in ("4k-dibr-gui", "1080p-gui-defaults"):
            rf = (srs.get(n) or {}).get("roofline")
            if rf:
                out["sub_records"][n]["w1_frac"] = rf.get("frac")
                out["sub_records"][n]["w1_avg_launch_ms"] = rf.get("avg_launch_ms")
    if srs and any(n in srs for n in ("4k-dav2b-dibr-f32x3", "4k-dav2b-dibr-fp16x2")):
        # the headline workload in the two OPT-IN float32-operand modes of the depth leg (split onto the 16-bit matrix cores, float32 accumulation; never `value`)
        out["opt_in_depth_modes"] = {k: (srs.get(n) or {}).get("value") for k, n in (("bf16x3", "4k-dav2b-dibr-f32x3"), ("fp16x2", "4k-dav2b-dibr-fp16x2"))}
    out["note"] = ("compact line; notes, stage tables and per-sub-record rooflines: " + (full_path or "the earlier stdout line `bench_full_record`") +
                   "; W1/E1 frac = SURVEY 8(d) algorithmic bytes / HIP-event launch time / 8 TB/s; both kernels are VALU-bound (valu_*)")
    if res.get("commit"):
        out["commit"] = res["commit"]
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LINE_LIMIT:   # never lose the measurement to the size of the line again: drop the optional parts, in this order
        for k in ("cpu_baseline_1080p", "roofline_chain", "note"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= COMPACT_LINE_LIMIT:
                break
    if len(line) > COMPACT_LINE_LIMIT and "sub_records" in out:
        out["sub_records"] = {n: {"value": r.get("value")} for n, r in out["sub_records"].items()}
    return out


def emit_record(res, full_path=None):
    """Print the full record on one stdout line (wrapped, so that it cannot be mistaken for the bench line), write it to `full_path`, then
    print the compact bench line LAST."""
    full_path = full_path or os.environ.get("VD3D_BENCH_FULL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    wrote = None
    try:
        os.makedirs(os.path.dirname(full_path), exist_ok=True)
        with open(full_path, "w") as f:
            json.dump(res, f, indent=1)
        wrote = os.path.relpath(full_path, ROOT)
    except Exception:
        pass
    sys.stderr.flush()
    print(json.dumps({"bench_full_record": res}, separators=(",", ":")), flush=True)
    line = json.dumps(compact_record(res, wrote), separators=(",", ":"))
    print(line, flush=True)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="measure ONLY this workload (profiling runs); default: the headline + the sub-records described above")
    ap.add_argument("--depth-dtype", default="f32", choices=("f32", "bf16", "f32x3", "f32h2"), help="depth-net precision of the measured workload (f32x3 / f32h2: float32 with the transformer blocks on the split-bf16 / 2 x fp16 MFMA kernels, opt-in) "
                    "(f32 = the reference's; bf16 is labelled reduced precision)")
    ap.add_argument("--batch", type=int, default=16, help="frames per step")
    ap.add_argument("--clip", type=int, default=32, help="distinct synthetic frames resident in HBM (cycled); default 2 x batch so that "
                    "consecutive steps render different frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true", help="headline only")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-stage HIP-event pass behind the timed region")
    ap.add_argument("--no-miopen-find", action="store_true", help="depth net: MIOpen's immediate-mode solver choice instead of find mode (find mode times every "
                    "convolution shape's solvers once per process, ~25 s inside the first warm-up step)")
    ap.add_argument("--sharded", action="store_true", help="use the chunk-sharding step protocol even at N=1 without pixel overlap (the default since round 4)")
    ap.add_argument("--per-frame", action="store_true", help="one vd3d_render_frame call per frame instead of the batched step protocol (N = 1, no pixel overlap)")
    ap.add_argument("--host-io", action="store_true", help="frames start in (pinned) host memory and muxed frames end there: "
                    "PCIe-inclusive rate through visiondepth3d_amd.frame_io.PinnedRing (not the contract's `value`)")
    ap.add_argument("--host-io-nv12", action="store_true", help="like --host-io with NV12 frames on the wire both ways (half the PCIe bytes; the colour "
                    "conversions run on the device: vd3d_nv12_to_bgr / vd3d_bgr_to_nv12)")
    ap.add_argument("--ring-depth", type=int, default=4, help="slots of the pinned staging ring of --host-io (H2D, render and D2H of consecutive steps overlap)")
    ap.add_argument("--pixel-overlap", dest="pixel_overlap", action="store_true", default=None,
                    help="two slot sets + vd3d_set_pixel_overlap: the pixel kernels of step i run on a second stream of the renderer while "
                    "the (latency-bound) measurement chain of step i+1 runs on the first (the default)")
    ap.add_argument("--no-pixel-overlap", dest="pixel_overlap", action="store_false")
    ap.add_argument("--pix-streams", type=int, default=2, help="pixel streams of the renderer (vd3d_set_pixel_overlap(ctx, n)): consecutive frames' "
                    "k_shift / W1 / E1 go round-robin over n streams so that neighbouring frames' kernels share the CUs")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the DIBR chain on the depth net's stream instead of a private HIP stream (no cross-batch overlap)")
    ap.add_argument("--upscale-only", action="store_true", help="measure only the configs[4] sub-record (1080p depth + DIBR + Real-ESRGAN x4)")
    ap.add_argument("--chain-serial", action="store_true", help="configs[4] sub-record: depth, DIBR and the up-scale net on ONE stream (default: the up-scale "
                    "net of a batch on a second stream behind depth + DIBR of the next)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))   # `python bench.py --gpus N` starts its own N ranks (one per GPU)
    env = Env(args)
    if args.upscale_only:
        print(json.dumps(run_upscale_chain(env, args)), flush=True)
        return
    single = args.workload is not None
    wl = args.workload or HEADLINE
    prof = not args.no_profile
    head = run_workload(env, args, wl, args.steps, args.warmup, depth_dtype=args.depth_dtype, profile=prof)
    subs, roof_src = {}, head
    if env.rank == 0 and env.world == 1 and not single and not args.no_sub_records:
        nroof = max(2, math.ceil(200 / args.batch))   # >= 200 timed frames, >= 20 warm-up frames
        r4 = run_workload(env, args, "4k-dibr", nroof, max(6, math.ceil(20 / args.batch)), profile=prof)   # (round 6: six warm-up steps -- the sub-records follow the float32 headline and measured 3 - 8 % under a run of their own with two)
        r1e = run_workload(env, args, "1080p-dav2s-dibr", 10, 3, depth_dtype="f32", profile=prof)
        r1d = run_workload(env, args, "1080p-dibr", 24, 8, profile=prof)
        rbf = run_workload(env, args, HEADLINE, 10, 3, depth_dtype="bf16", profile=prof, isolated_pass=False)
        rdn = run_workload(env, args, "4k-dibr-sepdof", 12, 6, profile=prof, isolated_pass=False)
        rd3 = run_workload(env, args, "4k-dibr-dof3", 12, 6, profile=prof, isolated_pass=False)
        ran = run_workload(env, args, "4k-dibr-anaglyph", 12, 6, profile=prof, isolated_pass=False)
        try:
            rvr = run_workload(env, args, "4k-dibr-vr", 12, 6, profile=prof, isolated_pass=False)
        except Exception as e:   # a sub-record must never take the headline down
            rvr = None
            print(f"[bench] 4k-dibr-vr failed: {str(e)[:200]}", file=sys.stderr)
        rg1 = run_workload(env, args, "1080p-gui-defaults", 24, 8, profile=prof)
        rg4 = run_workload(env, args, "4k-dibr-gui", 16, 6, profile=prof)
        rhi = run_workload(env, args, "4k-dibr", 6, 2, profile=False, isolated_pass=False, host_io=True)   # host-I/O-included figure
        rhn = None
        try:
            rhn = run_workload(env, args, "4k-dibr", 6, 2, profile=False, isolated_pass=False, host_io="nv12")
            rhn["workload"] = "4k-dibr-hostio-nv12"
            rhn["desc"] = ("like 4k-dibr-hostio with NV12 frames on the wire both ways (12.4 MB up, 12.4 MB down per pair instead of 24.9 each): "
                           "vd3d_nv12_to_bgr in front of the step, vd3d_bgr_to_nv12 behind it, both on the device; never `value`")
        except Exception as e:
            print(f"[bench] 4k-dibr-hostio-nv12 failed: {str(e)[:200]}", file=sys.stderr)
        rhi["workload"] = "4k-dibr-hostio"
        rhi["desc"] = ("4K DIBR only with the frames starting in pinned host memory and the muxed frames copied back to pinned host memory "
                       "(frame_io.PinnedRing, three slots: H2D, render and D2H of consecutive steps overlap): the PCIe-inclusive rate, never `value`")
        subs = {"4k-dibr": (r4, None), "1080p-dav2s-dibr": (r1e, None), "1080p-dibr": (r1d, None), "4k-dav2b-dibr-bf16": (rbf, None),
                "4k-dibr-sepdof": (rdn, None), "4k-dibr-dof3": (rd3, None), "4k-dibr-anaglyph": (ran, None), "4k-dibr-hostio": (rhi, None),
                "1080p-gui-defaults": (rg1, None), "4k-dibr-gui": (rg4, None)}
        if rvr is not None:
            subs["4k-dibr-vr"] = (rvr, None)
        if rhn is not None:
            subs["4k-dibr-hostio-nv12"] = (rhn, None)
        roof_src = r4
        try:
            up_rec = run_upscale_chain(env, args)
        except Exception as e:   # a sub-record must never take the headline down
            up_rec = {"workload": "1080p-dav2s-dibr-esrgan4k", "error": str(e)[:300]}
        # round 6, LAST (two full runs of this file with them in front of the DIBR-only sub-records measured those 8 - 10 % lower, the host-bound ones more; a probe
        # that alternates them with `4k-dibr-gui` in one process does not reproduce it -- tools/probe_after_x3.py -- so they simply go where they can disturb nothing):
        # the headline workload with the transformer blocks on the library's split-operand MFMA kernels (float32 operands, float32 accumulation; opt-in, never `value`)
        for name, dd in (("4k-dav2b-dibr-f32x3", "f32x3"), ("4k-dav2b-dibr-fp16x2", "f32h2")):
            try:
                subs[name] = (run_workload(env, args, HEADLINE, 8, 4, depth_dtype=dd, profile=prof, isolated_pass=False), None)
            except Exception as e:
                print(f"[bench] {name} failed: {str(e)[:200]}", file=sys.stderr)

    if env.rank == 0:
        copy_gbs = copy_yardstick(env)
        sh, sw, model_name, desc = WORKLOADS[wl]
        value = head["frames_total"] / head["dt"]
        reduced = model_name is not None and args.depth_dtype == "bf16"
        res = {
            "metric": "stereo-pairs/sec end-to-end (depth+warp+fill+mux)",
            "value": round(value, 3), "unit": "stereo-pairs/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["dt"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ({"f32x3": "f32 (bf16x3 split MFMA, f32 accumulate)", "f32h2": "f32 operands as 2 x fp16 (22 bits), MFMA, f32 accumulate"}.get(args.depth_dtype, "f32")
                      if model_name else "f32") if not reduced
                     else "f32 DIBR + bf16 depth net (REDUCED precision vs the reference's float32)",
            "data": "synthetic (procedural frames+depth resident in HBM, deterministic synthetic depth-net weights)" if not args.host_io else
                    "synthetic; frames start in pinned host memory and muxed frames are copied back to pinned host memory (PCIe-inclusive run)",
            "config": {"workload": wl, "description": desc, "frame": f"{sw}x{sh}", "format": "Half-SBS",
                       "frames_per_step": args.batch, "depth_model": model_name,
                       "depth_net_dtype": ({"f32": "float32 (the reference's precision)", "bf16": "bfloat16",
                                            "f32x3": "float32 via split-bf16 linears + attention (opt-in)",
                                            "f32h2": "float32 via 2 x fp16 linears + attention (opt-in)"}[args.depth_dtype]
                                           if model_name else None),
                       "arithmetic": "u8 in/out, float32 DIBR kernels, float64 scalar trackers",
                       "depth_net_library_selection": head.get("lib_sel"),
                       "pixel_overlap": head["pix_ov"], "pixel_streams": head.get("pix_streams"),
                       "sharding": "contiguous frame chunks per rank; scalar records all-gathered and trackers replayed on every rank "
                                   "(bit-identical to 1 GPU)" if env.world > 1 else None,
                       "rccl_ranks": env.world if env.world > 1 else None,
                       "comm_per_step_per_rank": head.get("shard_bytes") if env.world > 1 else None,
                       "distinct_frames_per_rank": head.get("clip"), "clip_frames_all_ranks": head.get("clip_frames_global"),
                       "clip_layout": "one clip, contiguous chunks of frames_per_step per rank and step" if env.world > 1 else None,
                       "p1_chain_wait_ms_per_step": head.get("p1_wait_ms") if env.world > 1 else None,
                       "p1_chain_wait_ms_per_step_min_max": ([_P1_WAIT_MIN[0], head.get("p1_wait_ms")] if env.world > 1 else None),
                       "rank_pids": env.rank_pids if env.world > 1 else None,
                       "params": "render_cli.py defaults + dof_strength 2.0; DOF levels in the reference's dense convolution order (parity mode)"},
        }
        hr = rooflines(head, copy_gbs)
        if roof_src is not head:   # the section-8(d) roofline run: 4K, DIBR only, >= 200 timed frames
            rr = rooflines(roof_src, copy_gbs)
            for k in ("roofline", "roofline_e1", "roofline_chain"):
                if k in rr:
                    rr[k]["measured_on"] = f"sub-record 4k-dibr ({roof_src['frames_total']} timed frames, DIBR only)"
                    if k in hr and "avg_launch_ms" in hr[k]:
                        rr[k]["in_headline_avg_launch_ms"] = hr[k]["avg_launch_ms"]   # same kernel while the depth net shares the CUs
                    res[k] = rr[k]
            if "roofline_depthnet" in hr:
                res["roofline_depthnet"] = hr["roofline_depthnet"]
        else:
            res.update(hr)
        res["stage_ms"] = head["stage_ms"]
        if subs:
            sr = {}
            for name, (rs, _) in subs.items():
                extra = {"stage_ms": rs["stage_ms"]}
                rf = rooflines(rs, copy_gbs, pmc_workload=None)
                if "roofline_depthnet" in rf:
                    extra["roofline_depthnet"] = rf["roofline_depthnet"]
                if name in ("1080p-gui-defaults", "4k-dibr-gui"):   # W1 where north_star's HBM question is meaningful: no feathering (round 5)
                    rg = rooflines(rs, copy_gbs, pmc_workload=name)
                    for k in ("roofline", "roofline_e1"):
                        if k in rg:
                            extra[k] = rg[k]
                    extra["geometry"] = {"warp": rs["warp"], "eye": rs["eye"], "out": rs["out"]}
                if name == "4k-dibr-sepdof":
                    extra["note"] = ("opt-in fast mode of the finishing stage (DESIGN.md section 2); every other record, the headline included, "
                                     "runs the dense association that matches the reference's CPU result exactly")
                if name == "4k-dibr-hostio-nv12":
                    bpf = rs["sh"] * rs["sw"] * 3   # 1.5 bytes per pixel each way
                    extra["pcie_bytes_per_frame"] = bpf
                    extra["pcie_GBs_each_way"] = round(bpf / 2 * rs["frames_total"] / rs["dt"] / 1e9, 2)
                    extra["note"] = "host-I/O-inclusive with NV12 on the wire (BT.601 limited range, 4:2:0: NOT the reference's bgr24 bytes -- an opt-in wire format)"
                if name == "4k-dibr-hostio":
                    bpf = rs["sh"] * rs["sw"] * 3 * 2   # one source frame up, one muxed Half-SBS frame (same size) down; the float32 depth planes stay in HBM
                    extra["pcie_bytes_per_frame"] = bpf
                    extra["pcie_GBs_each_way"] = round(bpf / 2 * rs["frames_total"] / rs["dt"] / 1e9, 2)
                    extra["note"] = "host-I/O-inclusive (PCIe Gen5 x16, 63 GB/s spec each way); depth planes precomputed and resident, like `4k-dibr`"
                if name == "4k-dav2b-dibr-f32x3":
                    extra["note"] = ("same workload as the headline with the four linears of every transformer block on vd3d_gemm_x3: every float32 operand split "
                                     "exactly into three bf16 terms, six bf16 MFMA products per MAC, float32 accumulation, exact GELU in fc1's epilogue -- "
                                     "float32-faithful (tests/test_hip_gemm.py: vs float64, and the float32 leg's own bar against the stock graph), opt-in; the "
                                     "headline stays pure float32 (hipBLASLt)")
                if name == "4k-dav2b-dibr-fp16x2":
                    extra["note"] = ("same workload as the headline with the transformer blocks on the library's GEMM / attention kernels in their fp16x2 form: every "
                                     "operand as two fp16 terms (22 significant bits, round to nearest), three MFMA products per MAC, float32 accumulation -- half the "
                                     "matrix work of bf16x3; measured against float64 beside hipBLASLt / AOTriton in tests/test_hip_gemm.py, and on the uint8 depth plane "
                                     "against the stock float32 graph; opt-in, never `value`")
                if name == "4k-dav2b-dibr-bf16":
                    extra["note"] = ("same workload as the headline with the depth net in bfloat16: NOT like-for-like with the reference "
                                     "(float32); its uint8 depth-plane deviation is measured by tests/test_hip_depth_e2e.py")
                sr[name] = sub_record(rs, extra)
            sr[up_rec["workload"]] = up_rec
            res["sub_records"] = sr
        if env.world == 1 and not args.no_cpu_baseline:
            # cpu_baseline = the oracle on ALL host cores (one independent clip per process: the CPU path's whole-socket rate, the fair
            # comparison for a whole GPU); the 1-core figure of the same port rides along as `single_core`
            def both(h, w, frames_per_core, budget, maxf, timeout_s):
                one = cpu_baseline(h, w, seconds_budget=budget, max_frames=maxf)
                try:
                    allc = cpu_baseline_allcores(h, w, frames_per_core, timeout_s=timeout_s)
                except Exception as e:
                    allc = {"error": str(e)[:200]}
                if "error" in allc:
                    one["all_cores"] = allc
                    return one
                allc["single_core"] = one
                return allc
            cb = both(sh, sw, 1 if sh > 1080 else 2, 12.0, 8 if sh > 1080 else 30, 120.0)
            if subs and wl == HEADLINE:
                cb["gpu_same_work"] = {"workload": "4k-dibr", "value": sub_record(subs["4k-dibr"][0])["value"], "unit": "stereo-pairs/s",
                                       "note": "DIBR chain only on one MI355X: the same work as this CPU figure (`value` also holds the depth net)"}
            res["cpu_baseline"] = cb
            if subs:
                c1 = both(1080, 1920, 2, 8.0, 30, 90.0)
                c1["gpu_same_work"] = {"workload": "1080p-dibr", "value": sub_record(subs["1080p-dibr"][0])["value"], "unit": "stereo-pairs/s"}
                res["cpu_baseline_1080p"] = c1
                try:
                    res["cpu_depth_net"] = cpu_depth_net(model_name, sh, sw) if model_name else None
                except Exception as e:
                    res["cpu_depth_net"] = {"error": str(e)[:200]}
        res["commit"] = _git_commit()
    if env.world > 1:
        env.dist.destroy_process_group()   # before the line: nothing may print after it
    if env.rank == 0:
        emit_record(res)


if __name__ == "__main__":
    if len(sys.argv) == 6 and sys.argv[1] == "--cpu-worker":   # one core of cpu_baseline_allcores (no torch, no GPU)
        print(_cpu_worker(tuple(int(v) for v in sys.argv[2:6])))
    else:
        main()
