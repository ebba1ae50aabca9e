#!/bin/bash
# round 5, call 2: where does the body convolution spend its time?  phase stamps of both launch policies
O=gpurun_out/r05c2; mkdir -p $O
VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 120 tools/gpu_ab.bin conv 540 960 2>&1 | tee $O/conv_stamps.log
