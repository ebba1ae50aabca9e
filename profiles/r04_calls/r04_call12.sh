# round-4 GPU call 12: E1 tile heights 22 / 14 at 80 VGPRs (A/B libs), chain workgroup divisor and pixel-stream count re-tuned on the final kernels
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c12; mkdir -p $O
cd /tmp
for v in th22 th14; do
  rm -rf $O/kt
  VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 2 0:16:8:32 > $O/kt_$v.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 18 > $O/kt_$v.md 2>&1; rm -rf $O/kt
  echo "== $v"; grep -E "k_finish" $O/kt_$v.md | awk -F'|' '{printf "%-40s calls %s avg %s min %s vgpr %s lds %s\n", substr($2,1,40), $3, $5, $6, $9, $12}'
done
cd $R
timeout 400 python tools/probe_step.py --clip 8 2:16:8:32 2:16:4:32 2:16:16:32 3:16:8:32 1:16:8:32 2:8:8:32 > $O/probe.log 2>&1; tail -6 $O/probe.log
for v in th22 th14; do VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_$v.so timeout 200 python tools/probe_step.py --clip 8 2:16:8:32 > $O/probe_$v.log 2>&1; echo "$v: $(tail -1 $O/probe_$v.log)"; done
