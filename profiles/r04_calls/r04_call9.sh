# round-4 GPU call 9: W1 builds only the Hh chunks a tile's shifts reach: parity tests that touch the warp (incl. large shifts), kernel trace, throughput
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c9; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_edge_cases.py tests/test_hip_widen.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
rm -rf $O/kt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 4 0:16:8:32 > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 18 > $O/kt.md 2>&1; rm -rf $O/kt
grep -E "k_chain|k_shift|k_warp|k_finish|k_e2w" $O/kt.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s vgpr %s lds %s grid %s\n", substr($2,1,40), $3, $5, $6, $9, $12, $14}'
cd $R
timeout 300 python tools/probe_step.py --clip 8 --check 2:16:8:32 0:16:8:32 > $O/probe.log 2>&1; tail -3 $O/probe.log
