# round-3 GPU call 1: micro-benchmarks, FETCH_SIZE calibration, XCD band order A/B, f32 GEMM tuning, steady-state headline trace
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c1; mkdir -p $O
cd /tmp
( timeout 120 $R/tools/ubench_valu.bin > $O/ubench.log 2>&1 )
( rm -rf $O/pmc_ub; timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_ub -o p -- $R/tools/ubench_valu.bin > /dev/null 2>&1;
  python $R/tools/pmc_summary.py $(find $O/pmc_ub -name "*_results.db") > $O/fetch_calib.md 2>&1; rm -rf $O/pmc_ub )
DIBR="python $R/bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline"
for x in 0 1; do
  VD3D_XCD_ORDER=$x timeout 300 $DIBR > $O/dibr_xcd$x.json 2> $O/dibr_xcd$x.err
  rm -rf $O/p_f$x; VD3D_XCD_ORDER=$x timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p_f$x -o p -- $DIBR --no-profile > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find $O/p_f$x -name "*_results.db") > $O/fetch_xcd$x.md 2>&1; rm -rf $O/p_f$x
done
( cd $R && timeout 600 python tools/tune_gemm.py bench > $O/gemm_default.log 2>&1;
  timeout 900 python tools/tune_gemm.py tune $O/tunableop_f32.csv > $O/gemm_tune.log 2>&1;
  CSV=$(ls $O/tunableop_f32*.csv | head -1); timeout 300 python tools/tune_gemm.py bench $CSV > $O/gemm_tuned.log 2>&1 )
rm -rf $O/t_head; timeout 600 rocprofv3 --kernel-trace -d $O/t_head -o p -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-sub-records --no-profile > $O/head.json 2> $O/head.err
DB=$(find $O/t_head -name "*_results.db" | head -1)
python $R/tools/steady_state.py $DB 4 70 > $O/head_steady.txt 2>&1
rm -rf $O/t_head
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
ls -la $O
