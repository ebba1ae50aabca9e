# round-4 GPU call 8: pass B as candidate lists, W1 Hh batches of 14 rows (A/B: 6), k_e2w row table, format_3d_output / INTER_LINEAR: full GPU suite,
# kernel trace, throughput main vs the WF_HB=6 build
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c8; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
rm -rf $O/kt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 4 0:16:8:32 > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 18 > $O/kt.md 2>&1; rm -rf $O/kt
grep -E "k_chain|k_shift|k_warp|k_finish|k_e2w|fillBuffer|copyBuffer" $O/kt.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s vgpr %s lds %s grid %s\n", substr($2,1,40), $3, $5, $6, $9, $12, $14}'
rm -rf $O/kt
VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_hb6.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 4 0:16:8:32 > $O/kt_hb6.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 18 > $O/kt_hb6.md 2>&1; rm -rf $O/kt
echo "== hb6"; grep -E "k_warp|k_e2w" $O/kt_hb6.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s\n", substr($2,1,40), $3, $5, $6}'
cd $R
timeout 300 python tools/probe_step.py --clip 8 --check 2:16:8:32 0:16:8:32 > $O/probe.log 2>&1; tail -3 $O/probe.log
