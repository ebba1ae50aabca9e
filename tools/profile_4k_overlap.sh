export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/v_4k
rocprofv3 --kernel-trace --stats -d $O/v_4k -o p -- python bench.py --workload 4k-dibr --steps 10 --warmup 3 --no-cpu-baseline > $O/v_4k.log 2>&1
DB=$(find $O/v_4k -name "*_results.db" | head -1)
python tools/rocpd_summary.py $DB > $O/v_4k_kernel_stats.md
rm -f $DB
python bench.py --workload 4k-dibr --no-cpu-baseline 2>/dev/null | tail -1 > $O/v_bench_4k-dibr.json
python -c "
import json; d=json.load(open('$O/v_bench_4k-dibr.json')); rf=d['roofline']; print(d['value'], rf['frac'], rf.get("in_step_frac"), rf.get('avg_launch_ms'), rf.get("in_step_avg_launch_ms"), d['stage_ms'])"
