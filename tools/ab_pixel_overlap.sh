#!/bin/bash
# A/B of the overlapped pixel passes (vd3d_set_pixel_overlap) on one GPU box: same box, back to back.
#   gpurun --timeout 500 -- 'bash tools/ab_pixel_overlap.sh'
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rf=d.get('roofline',{})
print('$*', d['value'], d['ms_per_step'], rf.get('frac'), rf.get("in_step_frac"), d['config'].get('pixel_overlap'))"; }
for w in 4k-dibr 1080p-dibr 1080p-dav2s-dibr; do
  run --workload $w --no-pixel-overlap
  run --workload $w
done
