"""`python bench.py --gpus N` starts its own N ranks.  Two checks that need no GPU:

* the product command line (`--backend hip`, the default) with --gpus 2 on a machine without GPUs: the launcher starts two ranks, they
  rendezvous (gloo) and prove it, rank 0 prints one JSON line naming the problem, exit code 0 -- there is still no CPU fallback;
* the test driver `tests/bench_oracle_gloo.py` (round 5: it lived inside bench.py as `--backend oracle-gloo` before): bench.py's OWN launcher
  (`bench.self_launch`), `Env`, fence and record helpers, the clip layout (ONE clip cut into contiguous chunks) and the step protocol
  (sharded.ChunkSharder: point-to-point plane hand-off + two all-gathers) with the CPU oracle as the sharder's backend: the frames rendered by
  two -- and by eight -- ranks must carry the same checksums as the same clip rendered by one rank;
* bench.py itself no longer imports anything under oracle/ except in its `cpu_baseline` leg.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, timeout=600, script="bench.py"):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, script), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    return pr, (json.loads(lines[-1]) if lines else None)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour of the launcher")
def test_gpus2_without_gpu_starts_two_ranks_and_stops_cleanly():
    pr, rec = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert rec is not None and rec["ranks_started"] == 2 and rec["n_gpus"] == 2 and len(set(rec["rank_pids"])) == 2
    assert "no GPU" in rec["error"] and "value" not in rec


DRIVER = os.path.join("tests", "bench_oracle_gloo.py")


def test_gpus2_oracle_gloo_equals_one_rank():
    pr2, r2 = _run("--gpus", "2", "--steps", "2", "--warmup", "1", script=DRIVER)
    assert pr2.returncode == 0, pr2.stderr[-2000:]
    pr1, r1 = _run("--gpus", "1", "--steps", "4", "--warmup", "2", script=DRIVER)
    assert pr1.returncode == 0, pr1.stderr[-2000:]
    assert r2["n_gpus"] == 2 and r2["config"]["gloo_ranks"] == 2 and r1["n_gpus"] == 1
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert k in r2
    c = r2["config"]["comm_per_step_per_rank"]
    assert c["collectives_per_step"] == 2 and c["p2p_plane_bytes_sent"] == 36 * 64 * 4 and c["allgather_bytes_received"] == 4 * 40
    assert r2["config"]["p1_chain_wait_ms_per_step"] is not None
    lo, hi = r2["config"]["p1_chain_wait_ms_per_step_min_max"]
    assert 0.0 <= lo <= hi == r2["config"]["p1_chain_wait_ms_per_step"]
    assert len(set(r2["config"]["rank_pids"])) == 2
    # frames 4 .. 11 of ONE clip: rendered by two ranks in chunks of 2 == rendered by one rank
    assert sorted(map(int, r2["frame_checksums"])) == list(range(4, 12))
    assert r2["frame_checksums"] == r1["frame_checksums"]
    assert "TEST MODE" in r2["data"]


def test_gpus8_oracle_gloo_equals_one_rank():
    """World 8 through the launcher (the size the driver's SCALE run uses): eight processes, one step of 16 frames after a warm-up step."""
    pr8, r8 = _run("--gpus", "8", "--steps", "1", "--warmup", "1", script=DRIVER, timeout=900)
    assert pr8.returncode == 0, pr8.stderr[-2000:]
    pr1, r1 = _run("--gpus", "1", "--steps", "8", "--warmup", "8", script=DRIVER, timeout=900)
    assert pr1.returncode == 0, pr1.stderr[-2000:]
    assert r8["n_gpus"] == 8 and len(set(r8["config"]["rank_pids"])) == 8
    assert sorted(map(int, r8["frame_checksums"])) == list(range(16, 32))
    assert r8["frame_checksums"] == r1["frame_checksums"]
    assert r8["config"]["comm_per_step_per_rank"]["allgather_bytes_received"] == 16 * 40


def test_bench_py_keeps_the_oracle_out_of_its_multi_rank_paths():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "oracle_chunk" not in src and "--backend" not in src
    body = src[src.index("def run_workload"):]
    assert "from oracle" not in body.split("def main()")[0]           # only cpu_baseline / _cpu_worker (above run_workload) import oracle/
