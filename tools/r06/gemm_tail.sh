# round 6: the GEMM's tail pass (split-K slices for an XCD's short last round) against whole tiles; usage: bash tools/r06/gemm_tail.sh [quick]
python -m pytest tests/test_hip_gemm.py -m gpu -x -q -k "tail or faithful" 2>&1 | tail -2
for lib in "" notail; do
  L=""; [ -n "$lib" ] && L="VD3D_LIB_PATH=$PWD/visiondepth3d_amd/ab/libvd3d_hip_$lib.so"
  for m in bf16x3 fp16x2; do echo "== ${lib:-tail} $m"; env $L X3_MODE=$m python tools/probe_gemm_x3.py 21920 2>&1 | grep -v amdgpu.ids | cut -c1-60 | tail -7; done
  [ "$1" = quick ] && continue
  for dt in f32x3 f32h2; do env $L python bench.py --depth-dtype $dt --steps 10 --warmup 4 --no-cpu-baseline --no-sub-records --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('${lib:-tail}', '$dt', r['value'], r['ms_per_step'])"; done
done
