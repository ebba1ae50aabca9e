# round-4 GPU call 4: k_shift_e2 + W1 with the precomputed mask (tile heights 32 / 16): parity tests that touch the warp, kernel traces, throughput;
# pass-B kernels with their hits muted (VD3D_DBG=16: is it the global atomics?)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c4; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_edge_cases.py tests/test_hip_widen.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
for cfg in "0 32" "0 16" "16 32"; do
  set -- $cfg
  rm -rf $O/kt
  VD3D_DBG=$1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 4 0:16:8:$2 > $O/kt_dbg$1_th$2.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 16 > $O/kt_dbg$1_th$2.md 2>&1; rm -rf $O/kt
  echo "== dbg $1 w1 th $2"; grep -E "k_chain|k_shift|k_warp|k_finish" $O/kt_dbg$1_th$2.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s vgpr %s lds %s grid %s\n", substr($2,1,40), $3, $5, $6, $9, $12, $14}'
done
cd $R
timeout 300 python tools/probe_step.py --clip 8 --check 2:16:8:32 2:16:8:16 3:16:8:16 0:16:8:16 > $O/probe.log 2>&1; tail -6 $O/probe.log
