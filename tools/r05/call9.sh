#!/bin/bash
# round 5, call 9: persistent fused finishing kernel (LDS-DMA prefetch) vs the one-tile kernel, byte for byte + time
O=gpurun_out/r05c9; mkdir -p $O
for fmt in half full interlaced anaglyph; do timeout 120 tools/gpu_ab.bin finish 2160 3840 $fmt 2>&1 | grep -v library; done | tee $O/finish_ab.log
timeout 120 tools/gpu_ab.bin finish 1080 1920 half 2>&1 | grep -v library | tee -a $O/finish_ab.log
timeout 120 tools/gpu_ab.bin finish 1000 1900 full 2>&1 | grep -v library | tee -a $O/finish_ab.log
echo "=== dword-plane DMA build"
VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_dma4.so timeout 120 tools/gpu_ab.bin finish 2160 3840 half 2>&1 | grep -v library | tee -a $O/finish_ab.log
