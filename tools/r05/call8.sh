#!/bin/bash
O=gpurun_out/r05c8; mkdir -p $O
for wd in 2 4; do
for sz in "135 240" "270 480" "540 960" "720 1280" "1080 1920"; do
  echo "=== CV_WD=$wd $sz"
  VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_wd$wd.so timeout 60 tools/gpu_ab.bin conv $sz 2>&1 | grep -v library
done; done | tee $O/conv_sizes.log
