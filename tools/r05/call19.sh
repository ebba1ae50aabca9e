#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c19; mkdir -p $O
{ echo "=== FF_XG=2 (default)"; timeout 60 tools/gpu_ab.bin finish 2160 3840 half 2>&1 | grep "mode 0";
  for g in 1 4; do echo "=== FF_XG=$g"; VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_xg$g.so timeout 60 tools/gpu_ab.bin finish 2160 3840 half 2>&1 | grep "mode 0"; done; } | tee $O/e1_xg.log
for b in 8 16 24 32; do
  timeout 300 python bench.py --no-sub-records --no-cpu-baseline --steps $((192 / b)) --warmup $((64 / b)) --batch $b --clip $((2 * b)) 2>$O/b.err | tail -1 > $O/b.json
  python -c "
import json
try:
    d=json.load(open('$O/b.json')); print('batch $b', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')
except Exception as e: print('batch $b failed', e); print(open('$O/b.err').read()[-500:])"
done | tee $O/headline_batch.log
