#!/bin/bash
# Round-end verification on one GPU box: full GPU test suite, smoke(), the default bench line (saved for profiles/rNN_bench_default.json).
#   gpurun --timeout 1500 -- 'bash tools/final_verify.sh'
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
T0=$(date +%s); python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json; echo "bench.py wall: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench_default.json"))
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("in_step_frac"), d["cpu_baseline"]["value"])
print("roofline:", {k: d["roofline"].get(k) for k in ("avg_launch_ms", "rocprof_avg_launch_ms", "traffic", "traffic_over_algorithmic", "valu_frac_of_spec", "traffic_taken_at_commit")})
for k, s in d.get("sub_records", {}).items():
    print("  ", k, s.get("value"), s.get("error"), (s.get("roofline") or {}).get("frac"), (s.get("roofline") or {}).get("traffic_over_algorithmic"))
PY
