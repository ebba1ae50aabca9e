# round-4 GPU call 21: k_sharp_fit (the fused kernel's epilogue behind the unfused DOF kernels): parity on every route, dof3 timing + kernel trace
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c21; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_edge_cases.py tests/test_hip_widen.py tests/test_hip_parity.py tests/test_hip_fuzz.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
timeout 300 python bench.py --workload 4k-dibr-dof3 --steps 4 --warmup 2 --no-cpu-baseline 2>$O/dof3.err | tail -1 > $O/dof3.json
python -c "
import json; d=json.load(open('$O/dof3.json')); print('4k-dibr-dof3', d['value'], d['ms_per_step'], d.get('stage_ms'))"
cd /tmp
rm -rf $O/t; timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o p -- python $R/bench.py --workload 4k-dibr-dof3 --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap > $O/t.log 2>&1
DB=$(find $O/t -name "*_results.db" | head -1)
python $R/tools/rocpd_summary.py $DB 14 > $O/r04_4k_dibr_dof3_kernel_stats.md; rm -rf $O/t
head -12 $O/r04_4k_dibr_dof3_kernel_stats.md | cut -c1-200
