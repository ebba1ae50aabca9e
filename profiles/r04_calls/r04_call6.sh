# round-4 GPU call 6: E1 with one barrier less + anaglyph epilogue, batched render_pairs: full GPU suite; kernel trace + SQ / FETCH / WRITE PMC
# passes of the batched 4K step (every kernel alone: pixel streams off); throughput incl. anaglyph and dof 3.0
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c6; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
rm -rf $O/kt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 4 0:16:8:32 > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 16 > $O/kt.md 2>&1; rm -rf $O/kt
grep -E "k_chain|k_shift|k_warp|k_finish|k_e2w" $O/kt.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s vgpr %s lds %s grid %s\n", substr($2,1,40), $3, $5, $6, $9, $12, $14}'
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  n=$(echo $c | cut -d" " -f1)
  rm -rf $O/p_$n; timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/p_$n -o p -- python $R/tools/probe_step.py --clip 4 --steps 2 0:16:8:32 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $(find $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_WAVES -name "*_results.db") > $O/pmc_step.md 2>&1
rm -rf $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_WAVES
cat $O/pmc_step.md | cut -c1-260 | head -20
cd $R
timeout 300 python tools/probe_step.py --clip 8 --check 2:16:8:32 > $O/probe.log 2>&1; tail -2 $O/probe.log
timeout 300 python tools/probe_step.py --clip 8 --fmt "Red-Cyan Anaglyph" 2:16:8:32 > $O/probe_ana.log 2>&1; tail -1 $O/probe_ana.log
timeout 300 python tools/probe_step.py --clip 8 --dof 3.0 2:16:8:32 > $O/probe_dof3.log 2>&1; tail -1 $O/probe_dof3.log
