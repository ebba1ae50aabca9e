#!/bin/bash
# W1 without feathering: 64 x 16 tiles (five workgroups per CU) vs 64 x 32 (three)
export TMPDIR=/tmp
O=gpurun_out/r05c18; mkdir -p $O
for th in 32 16; do for wl in 4k-dibr-gui 1080p-gui-defaults 4k-dibr-gui-hsbs; do
  VD3D_TUNE="7:$th" timeout 200 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline 2>$O/b.err | tail -1 > $O/b.json
  python - <<PY
import json
try:
    d = json.load(open("$O/b.json")); r = d.get("roofline", {})
    print("TH=$th $wl", d["value"], "pairs/s  W1 seq ms", r.get("avg_launch_ms"), "frac", r.get("frac"), "in-step", r.get("in_step_avg_launch_ms"))
except Exception as e:
    print("TH=$th $wl failed", e); print(open("$O/b.err").read()[-600:])
PY
done; done | tee $O/w1_nofeather_th.log
