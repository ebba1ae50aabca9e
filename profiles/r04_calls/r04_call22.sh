# round-4 GPU call 22: k_sharp_mux<PRE> template: parity (fractional fits, VR, blank frames), VR / blank-path timing
export TMPDIR=/tmp
O=gpurun_out/c22; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_edge_cases.py tests/test_hip_widen.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
for f in 3 0; do
  VD3D_FUSED_FIT=$f timeout 300 python bench.py --workload 4k-dibr-vr --steps 4 --warmup 2 --no-cpu-baseline --no-pixel-overlap 2>/dev/null | tail -1 > $O/vr$f.json
  python -c "
import json; d=json.load(open('$O/vr$f.json')); print('4k-dibr-vr sequential VD3D_FUSED_FIT=$f', d['value'], d['stage_ms']['finish'])"
done
