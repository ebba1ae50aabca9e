# round-4 GPU call 16: GEMM tables of the DA-V2-Small shapes (1080p sub-record, configs[4] chain), steady-state kernel breakdown of the headline
# with the neck / head glue + library selection, 1080p end-to-end sub-record
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c16; mkdir -p $O
MODEL=depth-anything-v2-small FRAME=1080x1920 timeout 400 python tools/probe_net_tune.py gemm $O/small16.csv 16 > $O/small16.log 2>&1; tail -1 $O/small16.log
MODEL=depth-anything-v2-small FRAME=1080x1920 timeout 400 python tools/probe_net_tune.py gemm $O/small8.csv 8 > $O/small8.log 2>&1; tail -1 $O/small8.log
cd /tmp
CMD="python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-sub-records --no-profile"
rm -rf $O/t_head; timeout 900 rocprofv3 --kernel-trace -d $O/t_head -o p -- $CMD > $O/t_head.log 2>&1
DB=$(find $O/t_head -name "*_results.db" | head -1)
{ echo "# steady-state kernel breakdown of \`$CMD\` (last 4 steps of a rocprofv3 --kernel-trace run; tools/steady_state.py)"; echo; echo '```';
  python $R/tools/steady_state.py $DB 4 70; echo '```'; } > $O/r04_4k_dav2b_f32_steady.md; rm -rf $O/t_head
head -30 $O/r04_4k_dav2b_f32_steady.md
cd $R
timeout 600 python bench.py --workload 1080p-dav2s-dibr --no-cpu-baseline --no-sub-records 2>/dev/null | tail -1 > $O/bench_1080p.json
python -c "
import json; d=json.load(open('$O/bench_1080p.json')); print('1080p-dav2s', d['value'], d['ms_per_step'])"
