#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c20; mkdir -p $O
run() { # label, env
  env $2 timeout 200 python bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline 2>$O/b.err | tail -1 > $O/b.json
  python -c "
import json
try:
    d=json.load(open('$O/b.json')); r=d.get('roofline_e1',{}); print('$1', d['value'], 'pairs/s  E1 seq ms', r.get('avg_launch_ms'), 'in-step', r.get('in_step_avg_launch_ms'))
except Exception as e: print('$1 failed', e); print(open('$O/b.err').read()[-500:])"
}
{ run "XG=2 (default)" "A=1"
  run "XG=1" "VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_xg1.so"
  run "plain order (xcd off)" "VD3D_TUNE=8:0"
  run "XG=2 again" "A=1"
  run "XG=1 again" "VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_xg1.so"; } | tee $O/e1_order_bench.log
