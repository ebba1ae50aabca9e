#!/bin/bash
# round 5, call 4: weight prefetch distance of the body convolution (2 = rounds 2-4, 4, 6, 8), phase stamps
O=gpurun_out/r05c4; mkdir -p $O
for wd in 2 4 6 8; do
  echo "=== CV_WD=$wd"
  VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_wd$wd.so timeout 120 tools/gpu_ab.bin conv 540 960 2>&1 | grep -v "^library"
done | tee $O/conv_wd.log
