# round-4 GPU call 7: merged LDS histogram atomics, strip-based dense DOF kernel for > 9-tap Gaussians: full GPU suite, dof 3.0 / 5.0 timing,
# kernel trace of the dof 3.0 step, then the complete default bench line
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c7; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/probe_step.py --clip 8 2:16:8:32 0:16:8:32 > $O/probe.log 2>&1; tail -2 $O/probe.log
timeout 300 python tools/probe_step.py --clip 8 --dof 3.0 2:16:8:32 0:16:8:32 > $O/probe_dof3.log 2>&1; tail -2 $O/probe_dof3.log
timeout 300 python tools/probe_step.py --clip 8 --dof 5.0 2:16:8:32 > $O/probe_dof5.log 2>&1; tail -1 $O/probe_dof5.log
cd /tmp
rm -rf $O/kt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 2 --dof 3.0 0:16:8:32 > $O/kt_dof3.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 16 > $O/kt_dof3.md 2>&1; rm -rf $O/kt
grep -E "k_chain|k_shift|k_warp|k_finish|k_e2w|k_dof|k_sharp" $O/kt_dof3.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s vgpr %s lds %s grid %s\n", substr($2,1,40), $3, $5, $6, $9, $12, $14}'
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
