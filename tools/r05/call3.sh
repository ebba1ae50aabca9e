#!/bin/bash
# round 5, call 3: body convolution with bias / slope in LDS and one load batch; phase stamps again, then the production build
O=gpurun_out/r05c3; mkdir -p $O
VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 120 tools/gpu_ab.bin conv 540 960 2>&1 | tee $O/conv_stamps.log
timeout 120 tools/gpu_ab.bin conv 540 960 2>&1 | tee $O/conv.log
