# round-4 GPU call 19: k_dof_grade4 with compile-time tap counts and scalar-operand weights: parity (every tap count 3 .. 31), dof3 sub-record
export TMPDIR=/tmp
O=gpurun_out/c19; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_edge_cases.py tests/test_hip_widen.py tests/test_hip_parity.py tests/test_hip_fuzz.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
for w in 4k-dibr-dof3; do
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-pixel-overlap 2>$O/$w.err | tail -1 > $O/$w.json
  python -c "
import json; d=json.load(open('$O/$w.json')); print('$w sequential', d['value'], d['ms_per_step'], d.get('stage_ms'))"
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>$O/$w.err | tail -1 > $O/$w.ov.json
  python -c "
import json; d=json.load(open('$O/$w.ov.json')); print('$w', d['value'], d['ms_per_step'], d.get('stage_ms'))"
done
