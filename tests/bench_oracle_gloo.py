#!/usr/bin/env python3
"""TEST DRIVER (test infrastructure; imports oracle/ through tests/oracle_chunk.py): the multi-rank machinery of `bench.py` -- its rank
launcher (`bench.self_launch`), its `Env` (rendezvous, barrier + synchronise fence), the clip layout (ONE clip cut into contiguous chunks),
the chunked step protocol (visiondepth3d_amd.sharded.ChunkSharder: point-to-point plane hand-off + two record all-gathers), the
max-over-ranks timing and the record assembly -- with tests/oracle_chunk.OracleChunkBackend (the CPU oracle) as the sharder's backend over
gloo on a tiny clip.  Proves without hardware that `python bench.py --gpus N` runs N cooperating ranks whose frames equal the frames of one
rank; NOT a benchmark result and labelled as such.  Lived inside bench.py as `--backend oracle-gloo` until round 4 (the
oracle has no business inside the product benchmark file).

    python tests/bench_oracle_gloo.py --gpus 2 --steps 2 --warmup 1        (tests/test_bench_launcher.py)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for q in (ROOT, HERE):
    if q not in sys.path:
        sys.path.insert(0, q)

import bench  # noqa: E402  (the product benchmark: launcher, Env, record helpers)


def run_oracle_gloo(env, args, sh=72, sw=128):
    torch, dist = env.torch, env.dist
    from oracle_chunk import OracleChunkBackend          # test infrastructure (imports oracle/)
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.params import render_kwargs_to_params
    from visiondepth3d_amd.sharded import ChunkSharder
    torch.set_num_threads(1)
    B, world, rank = min(args.batch, 2), env.world, env.rank
    p = render_kwargs_to_params(sw, sh, output_height=sh, **bench.RENDER_KW)
    be = OracleChunkBackend(p)
    shr = ChunkSharder(be, rank, world, B)
    be.new_clip()
    nsteps = args.warmup + args.steps
    clip = {}
    for k in range(nsteps):
        for j in range(B):
            t = k * world * B + rank * B + j
            f, d = synth.synth_frame(t, sh, sw)
            clip[(k, j)] = (torch.from_numpy(f), torch.from_numpy(synth.depth_to_u8_bgr(d)[..., 0].copy()))
    sums = {}

    def step(k):
        fl = [clip[(k, j)][0] for j in range(B)]
        dl = [clip[(k, j)][1] for j in range(B)]
        return shr.render_step(fl, dl, first_step=(k == 0), more_steps=(k + 1 < nsteps))
    for k in range(args.warmup):
        step(k)
    env.fence()
    t0 = time.perf_counter()
    for k in range(args.warmup, nsteps):
        for j, o in enumerate(step(k)):   # a checksum per muxed frame, keyed by the frame's index in the clip
            sums[k * world * B + rank * B + j] = int(o.numpy().astype(np.uint64).sum()) * 31 + int(o.numpy()[::3, ::5].astype(np.uint64).sum())
    env.fence()
    dt = time.perf_counter() - t0
    parts = [sums]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        parts = [None] * world
        dist.all_gather_object(parts, sums)
    sums = {str(t): v for part in parts for t, v in part.items()}
    wait_max = bench._p1_wait(env, [shr])
    return {"metric": "stereo-pairs/sec end-to-end (depth+warp+fill+mux)", "value": round(world * args.steps * B / dt, 3), "unit": "stereo-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "TEST MODE: CPU oracle backend over gloo on a tiny synthetic clip -- launcher / protocol check, not a benchmark result",
            "config": {"workload": f"oracle-gloo-{sw}x{sh}", "frames_per_step": B, "rccl_ranks": None, "gloo_ranks": world,
                       "rank_pids": env.rank_pids,
                       "comm_per_step_per_rank": shr.bytes_per_step(), "clip_layout": "one clip, contiguous chunks of frames_per_step per rank and step",
                       "p1_chain_wait_ms_per_step": wait_max,
                       "p1_chain_wait_ms_per_step_min_max": ([bench._P1_WAIT_MIN[0], wait_max] if world > 1 else None)},
            "frame_checksums": dict(sorted(sums.items(), key=lambda kv: int(kv[0])))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(bench.self_launch(args.gpus, script=__file__))   # bench.py's own launcher, re-executing THIS file
    env = bench.Env(args, cpu_only=True)
    rec = run_oracle_gloo(env, args)
    if env.rank == 0:
        print(json.dumps(rec), flush=True)
    if env.world > 1:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
