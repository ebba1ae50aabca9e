#!/bin/bash
# round 5, call 1: persistent conv A/B, the exact no-feather warp, ADVICE r4 fixes, GUI-default workloads
export TMPDIR=/tmp
O=gpurun_out/r05c1; mkdir -p $O
timeout 120 tools/gpu_ab.bin conv 540 960 2>&1 | tee $O/conv_ab.log
timeout 60 tools/gpu_ab.bin conv 270 480 2>&1 | tee -a $O/conv_ab.log
timeout 60 tools/gpu_ab.bin conv 1080 1920 2>&1 | tee -a $O/conv_ab.log
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_hip_widen.py tests/test_hip_upscale.py -m gpu -x -q -k "feather_strength_zero or format_3d or upscale or conv or chunk_sharding" 2>&1 | tail -8 | tee $O/pytest.log
for wl in 4k-dibr-gui 1080p-gui-defaults 4k-dibr-gui-hsbs 4k-dibr; do
  timeout 200 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_$wl.err | tail -1 > $O/bench_$wl.json
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$wl.json"))
    print("$wl", d["value"], "stage_ms", d.get("stage_ms"))
    r = d.get("roofline", {})
    print("   W1", {k: r.get(k) for k in ("avg_launch_ms", "k_e2w_avg_launch_ms", "achieved", "frac", "algorithmic_bytes_per_launch", "in_step_avg_launch_ms")})
    r = d.get("roofline_e1", {})
    print("   E1", {k: r.get(k) for k in ("avg_launch_ms", "achieved", "frac", "in_step_avg_launch_ms")})
except Exception as e:
    print("$wl: no record", e); print(open("$O/bench_$wl.err").read()[-1500:])
PY
done
