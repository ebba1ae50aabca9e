# round 6: D1 of W1 by eye-res segments (one load + two ds_bpermute per Hh element pair) against the two-loads-per-chunk form; usage: bash tools/r06/w1_seg.sh OUTDIR
O=${1:-gpurun_out/w1seg}; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_hip_widen.py tests/test_hip_edge_cases.py tests/test_hip_fuzz.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for lib in "" seg0 sb7; do
  for wl in 4k-dibr 4k-dibr-gui 1080p-gui-defaults; do
    L=""; [ -n "$lib" ] && L="VD3D_LIB_PATH=$PWD/visiondepth3d_amd/ab/libvd3d_hip_$lib.so"
    env $L python bench.py --workload $wl --steps 30 --warmup 10 --no-cpu-baseline --no-profile --no-sub-records 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r.get('roofline',{}); print('${lib:-new}', '$wl', r['value'], r['ms_per_step'], ro.get('avg_launch_ms'), ro.get('frac'))"
  done
done | tee $O/ab.txt
