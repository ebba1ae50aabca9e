# round-3 GPU call 15: E1 level-set reduction by ballot instead of 512 same-address LDS atomics -- parity + phases + A/B
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c15; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_edge_cases.py tests/test_hip_widen.py -m gpu -q -x > $O/pytest.log 2>&1
tail -2 $O/pytest.log
VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 300 python tools/probe_phases.py > $O/phases.log 2>&1
grep -A6 "^E1 k_finish" $O/phases.log; grep -A1 "^E1 residency" $O/phases.log
cd /tmp
timeout 300 python $R/bench.py --workload 4k-dibr --steps 8 --warmup 2 --no-cpu-baseline --no-pixel-overlap > $O/seq.json 2>/dev/null
timeout 300 python $R/bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov.json 2>/dev/null
python - <<PY
import json
for m in ("seq", "ov"):
    j = json.loads(open("$O/%s.json" % m).read().strip().splitlines()[-1])
    print(m, round(j["value"], 1), "pairs/s", {k: round(x, 4) for k, x in j.get("stage_ms", {}).items() if k in ("finish", "w1", "warp", "select_dc", "frame")})
PY
