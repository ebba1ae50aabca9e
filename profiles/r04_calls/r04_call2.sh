# round-4 GPU call 2: per-kernel times of the batched chain (pix_streams 0 = everything on one stream) for workgroup divisors and with the
# histogram flush / adds disabled (timing probes), then a throughput sweep of larger divisors
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c2; mkdir -p $O
cd /tmp
for cfg in "0 1" "4 1" "12 1" "0 8" "0 32" "4 8"; do
  set -- $cfg
  rm -rf $O/kt
  VD3D_DBG=$1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/tools/probe_step.py --clip 4 --steps 4 0:16:$2 > $O/kt_dbg$1_div$2.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 14 > $O/kt_dbg$1_div$2.md 2>&1; rm -rf $O/kt
  echo "== dbg $1 div $2"; grep -E "k_chain|k_shift|k_warp|k_finish|fillBuffer|copyBuffer" $O/kt_dbg$1_div$2.md | cut -c1-60,100-175
done
cd $R
timeout 300 python tools/probe_step.py --clip 8 2:16:8 2:16:16 2:16:32 2:16:64 > $O/probe.log 2>&1; tail -5 $O/probe.log
