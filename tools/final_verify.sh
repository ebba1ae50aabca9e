#!/bin/bash
# Round-end verification on one GPU box: full GPU test suite, smoke(), default bench line, rocprofv3 kernel trace of the default
# bench command (summaries land in gpurun_out/, copy the .md files into profiles/), DIBR-only bench lines.
#   gpurun --timeout 560 -- 'bash tools/final_verify.sh'
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/v_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/v_smoke.log
python bench.py 2>$O/v_bench_default.err | tail -1 > $O/v_bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/v_bench_default.json"))
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("isolated_frac"), d["cpu_baseline"]["value"])
PY
rm -rf $O/v_1080
rocprofv3 --kernel-trace --stats -d $O/v_1080 -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/v_1080.log 2>&1
DB=$(find $O/v_1080 -name "*_results.db" | head -1)
python tools/rocpd_summary.py $DB > $O/v_1080_kernel_stats.md
python tools/steady_state.py $DB 5 40 > $O/v_1080_steady.txt
tail -1 $O/v_1080.log | cut -c1-300
rm -f $DB   # the table and the steady-state block are what gets committed
for w in 4k-dibr 1080p-dibr; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/v_bench_$w.json
  python -c "
import json; d=json.load(open('$O/v_bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stage_ms',{}).get('w1'), d.get('stage_ms',{}).get('finish'))"
done
