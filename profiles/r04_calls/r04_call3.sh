# round-4 GPU call 3: chain v2 (streaming ingest + K1a, 16-byte rect walks, hull pre-test) + E1 at 80 VGPRs (folded levels, no SLP): full GPU suite,
# kernel trace of the batched step, throughput of the main build vs an all-files -fno-slp-vectorize build
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
rm -rf $O/kt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 4 0:16:8 > $O/kt_main.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 16 > $O/kt_main.md 2>&1; rm -rf $O/kt
grep -E "k_chain|k_shift|k_warp|k_finish" $O/kt_main.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s vgpr %s lds %s grid %s\n", substr($2,1,40), $3, $5, $6, $9, $12, $14}'
cd $R
timeout 300 python tools/probe_step.py --clip 8 2:16:8 0:16:8 > $O/probe_main.log 2>&1; tail -2 $O/probe_main.log
VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_noslp.so timeout 300 python tools/probe_step.py --clip 8 2:16:8 0:16:8 > $O/probe_noslp.log 2>&1; tail -2 $O/probe_noslp.log
