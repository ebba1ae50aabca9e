# round-3 GPU call 6: headline A/B of the E1 geometry on ONE box (64x30 product build vs 64x14 vs the round-2 64x16 geometry), clip 16 vs 32, large-blur tests
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c6; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_edge_cases.py -m gpu -q > $O/pytest_edge.log 2>&1
HEAD="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-sub-records"
for rep in 1 2; do
  timeout 300 $HEAD > $O/head_th30_$rep.json 2>/dev/null
  VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_th14.so timeout 300 $HEAD > $O/head_th14_$rep.json 2>/dev/null
  VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_geo16.so timeout 300 $HEAD > $O/head_geo16_$rep.json 2>/dev/null
done
timeout 300 $HEAD --clip 16 > $O/head_th30_clip16.json 2>/dev/null
ls -la $O
