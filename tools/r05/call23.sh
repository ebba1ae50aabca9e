#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c23; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_torch_math.py tests/test_hip_fuzz.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_a.log
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "aten" 2>&1 | tail -15 | tee $O/pytest_b.log
