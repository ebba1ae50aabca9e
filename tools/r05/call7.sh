#!/bin/bash
# ablations of the 32 x 8 body convolution (timing only: the results are wrong by construction): which resource bounds the layer?
O=gpurun_out/r05c7; mkdir -p $O
for v in NOMFMA NOSTORE NOLOAD NOMEM NOLOADMFMA; do
  echo "=== $v"
  VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_dbg_$v.so timeout 60 tools/gpu_ab.bin conv 540 960 2>&1 | grep "mode 103\|one tile"
done | tee $O/conv_ablate.log
