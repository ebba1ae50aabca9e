# round-4 GPU call 17: fused finishing kernel in front of any fit (VR / fractional INTER_AREA): parity both routes, VR / dof3 / anaglyph sub-record timings
export TMPDIR=/tmp
O=gpurun_out/c17; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_widen.py tests/test_hip_edge_cases.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
for w in 4k-dibr-vr 4k-dibr; do
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>$O/$w.err | tail -1 > $O/$w.json
  python -c "
import json; d=json.load(open('$O/$w.json')); print('$w', d['value'], d['ms_per_step'], d.get('stage_ms'))"
done
