# per-kernel times of the sequential 4K DIBR frame (no stream overlap) with the chain's debug knobs:
#   VD3D_DBG=0 normal | 1 skip the last-workgroup scan + scalar stage | 3 also skip tickets / fences   (1 and 3 give garbage pixels: timing only)
export TMPDIR=/tmp
O=gpurun_out
for dbg in 0 1 3; do
  rm -rf $O/chain_$dbg
  VD3D_DBG=$dbg rocprofv3 --kernel-trace --stats -d $O/chain_$dbg -o p -- python bench.py --workload 4k-dibr --steps 4 --warmup 2 --no-cpu-baseline --no-pixel-overlap --no-profile > $O/chain_$dbg.log 2>&1
  DB=$(find $O/chain_$dbg -name "*_results.db" | head -1)
  python tools/rocpd_summary.py $DB 16 > $O/chain_${dbg}_kernel_stats.md
  rm -rf $O/chain_$dbg
done
