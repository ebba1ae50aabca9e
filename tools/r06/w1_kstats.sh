# per-kernel times of the DIBR-only workloads for the product library and A/B libraries; usage: bash tools/r06/w1_kstats.sh OUTDIR lib1 lib2 ... ("" = product)
export TMPDIR=/tmp
R=$PWD; O=$R/${1:-gpurun_out/w1k}; shift; mkdir -p $O; cd /tmp
for lib in "$@"; do
  for wl in ${WLS:-4k-dibr 4k-dibr-gui}; do
    L=""; [ "$lib" != "prod" ] && L="VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_$lib.so"
    rm -rf $O/t; env $L rocprofv3 --kernel-trace --stats -d $O/t -o p -- python $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap --no-sub-records > $O/log_${lib}_$wl.txt 2>&1
    DB=$(find $O/t -name "*_results.db" | head -1)
    echo "== $lib $wl"; python $R/tools/rocpd_summary.py $DB 12 | cut -c1-160
  done
done > $O/kstats.txt 2>&1
rm -rf $O/t
