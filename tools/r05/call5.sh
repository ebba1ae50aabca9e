#!/bin/bash
O=gpurun_out/r05c5; mkdir -p $O
VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 120 tools/gpu_ab.bin conv 540 960 2>&1 | grep "where\|skew  5\|skew  0" | tee $O/conv_where.log
