# rocprofv3 evidence for the MFMA conv kernel (run on the GPU box through gpurun; outputs under gpurun_out/profiles_conv/):
# kernel trace + three separate PMC passes over tools/probe_conv.py (never combined with trace domains other than --kernel-trace)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profiles_conv; mkdir -p $O
cd /tmp
CMD="python $R/tools/probe_conv.py"
rm -rf $O/t; rocprofv3 --kernel-trace --stats -d $O/t -o p -- $CMD > $O/t.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/t -name "*_results.db" | head -1) 12 > $O/r02_conv_kernel_stats.md; rm -rf $O/t
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | cut -d" " -f1)
  rm -rf $O/p_$n; rocprofv3 --kernel-trace --pmc $c -d $O/p_$n -o p -- $CMD > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $(find $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_VALU_MFMA_BUSY_CYCLES -name "*_results.db") > $O/r02_pmc_conv_raw.md
rm -rf $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_VALU_MFMA_BUSY_CYCLES
ls -la $O
