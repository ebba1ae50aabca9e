#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c21; mkdir -p $O
run() { # label, env, workload
  env $2 timeout 200 python bench.py --workload $3 --steps 13 --warmup 2 --no-cpu-baseline 2>$O/b.err | tail -1 > $O/b.json
  python -c "
import json
try:
    d=json.load(open('$O/b.json')); r=d.get('roofline',{}); e=d.get('roofline_e1',{}); print('$1 $3', d['value'], 'pairs/s  W1 seq ms', r.get('avg_launch_ms'), 'traffic x', r.get('traffic_over_algorithmic'), ' E1 seq ms', e.get('avg_launch_ms'))
except Exception as e: print('$1 failed', e); print(open('$O/b.err').read()[-500:])"
}
{ for wl in 4k-dibr 4k-dibr-gui; do
  run "W1 band order (default)" "A=1" $wl
  run "W1 row round-robin" "VD3D_TUNE=9:2" $wl
  run "W1 plain" "VD3D_TUNE=9:0" $wl
  run "W1 band again" "A=1" $wl
  run "W1 row rr again" "VD3D_TUNE=9:2" $wl
done; } | tee $O/w1_order_bench.log
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_widen.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
