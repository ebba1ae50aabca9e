# round-4 GPU call 5: k_shift + k_e2w + W1 with the precomputed mask, everything built with -fno-slp-vectorize: full GPU suite, kernel trace,
# throughput, phase stamps of W1 / E1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
for th in 32 16; do
  rm -rf $O/kt
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 4 0:16:8:$th > $O/kt_th$th.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 16 > $O/kt_th$th.md 2>&1; rm -rf $O/kt
  echo "== w1 th $th"; grep -E "k_chain|k_shift|k_warp|k_finish|k_e2w" $O/kt_th$th.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s vgpr %s lds %s grid %s\n", substr($2,1,40), $3, $5, $6, $9, $12, $14}'
done
cd $R
timeout 300 python tools/probe_step.py --clip 8 --check 2:16:8:32 2:16:8:16 3:16:8:32 0:16:8:32 > $O/probe.log 2>&1; tail -5 $O/probe.log
VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 200 python tools/probe_phases.py > $O/phases.log 2>&1; tail -22 $O/phases.log
