#!/bin/bash
# headline scheduling variants: DIBR overlapped with the depth net (default) vs on the net's stream; pixel streams
export TMPDIR=/tmp
O=gpurun_out/r05c13; mkdir -p $O
run() { # name, flags
  timeout 400 python bench.py --no-sub-records --no-cpu-baseline --steps 12 --warmup 4 $2 2>$O/$1.err | tail -1 > $O/$1.json
  python - <<PY
import json
try:
    d = json.load(open("$O/$1.json")); print("$1", d["value"], "pairs/s", d["ms_per_step"], "ms/step", {k: d["stage_ms"].get(k) for k in ("warp","finish","p1_own","p3_own")}, d.get("roofline_depthnet", {}).get("avg_batch_ms"))
except Exception as e:
    print("$1 failed", e); print(open("$O/$1.err").read()[-800:])
PY
}
run default ""
run no_overlap "--no-overlap"
run pix1 "--pix-streams 1"
run no_pixel_overlap "--no-pixel-overlap"
run no_overlap_no_pix "--no-overlap --no-pixel-overlap"
