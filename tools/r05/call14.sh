#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c14; mkdir -p $O
T0=$(date +%s); python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json; echo "bench.py wall: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05c14/bench_default.json"))
print("default:", d["value"], d["ms_per_step"], "W1 frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
for k, s in d.get("sub_records", {}).items():
    print("  ", k, s.get("value"), s.get("error"), s.get("pcie_GBs_each_way"), (s.get("roofline") or {}).get("frac"))
PY
tail -5 $O/bench_default.err
