#!/bin/bash
O=gpurun_out/r05c10; mkdir -p $O
for fmt in half full; do timeout 120 tools/gpu_ab.bin finish 2160 3840 $fmt 2>&1 | grep -v library; done | tee $O/finish_ab.log
timeout 120 tools/gpu_ab.bin finish 1080 1920 half 2>&1 | grep -v library | tee -a $O/finish_ab.log
timeout 120 tools/gpu_ab.bin finish 1000 1900 anaglyph 2>&1 | grep -v library | tee -a $O/finish_ab.log
