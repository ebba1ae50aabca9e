# round-4 GPU call 20: kernel trace of the sequential 4K dof_strength 3.0 run (what k_dof_grade4 / k_sharp_mux cost per launch)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c20; mkdir -p $O
cd /tmp
rm -rf $O/t; timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o p -- python $R/bench.py --workload 4k-dibr-dof3 --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap > $O/t.log 2>&1
DB=$(find $O/t -name "*_results.db" | head -1)
python $R/tools/rocpd_summary.py $DB 14 > $O/r04_4k_dibr_dof3_kernel_stats.md; rm -rf $O/t
cat $O/r04_4k_dibr_dof3_kernel_stats.md | head -30
