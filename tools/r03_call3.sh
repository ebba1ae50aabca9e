# round-3 GPU call 3: E1 bank-conflict fix, K3 split, full GPU tests, 4K DIBR timing + PMC + kernel trace, copy variants, headline
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c3; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
cd /tmp
( timeout 200 $R/tools/ubench_valu.bin 2>&1 | grep "^copy" > $O/copy.log )
DIBR="python $R/bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline"
timeout 300 $DIBR > $O/dibr.json 2> $O/dibr.err
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  n=$(echo $c | cut -d" " -f1)
  rm -rf $O/p_$n; timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/p_$n -o p -- $DIBR --no-profile > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $(find $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_WAVES -name "*_results.db") > $O/pmc_4k_dibr_raw.md 2>&1
rm -rf $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_WAVES
SEQ="python $R/bench.py --workload 4k-dibr --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap"
rm -rf $O/t_dibr; timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_dibr -o p -- $SEQ > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/t_dibr -name "*_results.db" | head -1) 30 > $O/4k_dibr_seq_kernel_stats.md 2>&1; rm -rf $O/t_dibr
timeout 400 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sub-records > $O/head.json 2> $O/head.err
ls -la $O
