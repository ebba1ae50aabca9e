# round-3 GPU call 12b: E1 64x26 + depth loads hoisted: parity, phases, A/B vs 64x30
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c12; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_edge_cases.py tests/test_hip_widen.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 300 python tools/probe_phases.py > $O/phases.log 2>&1
tail -9 $O/phases.log
cd /tmp
for v in base th30; do
  if [ $v = base ]; then unset VD3D_LIB_PATH; else export VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_$v.so; fi
  timeout 300 python $R/bench.py --workload 4k-dibr --steps 8 --warmup 2 --no-cpu-baseline --no-pixel-overlap > $O/seq_$v.json 2>/dev/null
  timeout 300 python $R/bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov_$v.json 2>/dev/null
done
unset VD3D_LIB_PATH
timeout 300 python $R/bench.py --workload 1080p-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov1080_base.json 2>/dev/null
python - <<PY
import json
for v, m in (("base", "seq"), ("base", "ov"), ("th30", "seq"), ("th30", "ov"), ("base", "ov1080")):
    try:
        j = json.loads(open("$O/%s_%s.json" % (m, v)).read().strip().splitlines()[-1])
        print(v, m, round(j["value"], 1), "pairs/s", {k: round(x, 4) for k, x in j.get("stage_ms", {}).items() if k in ("finish", "w1", "warp", "select_dc", "frame")})
    except Exception as e:
        print(v, m, "ERR", e)
PY
