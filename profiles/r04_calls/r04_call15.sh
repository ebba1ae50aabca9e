# round-4 GPU call 15: DPT neck / head glue kernels (parity tests + A/B), library selection combined, headline
export TMPDIR=/tmp
O=gpurun_out/c15; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_depth_e2e.py tests/test_hip_depthprep.py tests/test_hip_widen.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
T=visiondepth3d_amd/tuned/gemm_gfx950.csv
VD3D_NECK_GLUE=0 timeout 300 python tools/probe_net_tune.py use $T > $O/use_noglue.log 2>&1; tail -2 $O/use_noglue.log
timeout 300 python tools/probe_net_tune.py use $T > $O/use_glue.log 2>&1; tail -2 $O/use_glue.log
FIND=1 timeout 400 python tools/probe_net_tune.py use $T > $O/use_glue_find.log 2>&1; tail -3 $O/use_glue_find.log
timeout 600 python bench.py --no-sub-records --no-cpu-baseline 2>$O/bench_head.err | tail -1 > $O/bench_head.json
python -c "
import json; d=json.load(open('$O/bench_head.json')); print('headline', d['value'], d['ms_per_step'], d['config'].get('depth_net_library_selection'), d.get('roofline_depthnet',{}).get('avg_batch_ms'))"
