python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "overlapped or measure_replay" 2>&1 | tail -4
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rf=d.get('roofline',{})
print('$*', d['value'], d['ms_per_step'], rf.get('frac'), rf.get('isolated_frac'), d['config'].get('pixel_overlap'))"; }
run
run --emulate-world 8
run --emulate-world 8 --no-pixel-overlap
run --host-io
run --workload 4k-dav2b-dibr --steps 8 --warmup 3
run --workload 4k-dav2b-dibr --steps 8 --warmup 3 --no-pixel-overlap
