#!/bin/bash
O=gpurun_out/r05c6; mkdir -p $O
timeout 120 tools/gpu_ab.bin conv 540 960 2>&1 | tee $O/conv_split.log
timeout 120 tools/gpu_ab.bin conv 1080 1920 2>&1 | tee -a $O/conv_split.log
timeout 120 tools/gpu_ab.bin conv 135 240 2>&1 | tee -a $O/conv_split.log
