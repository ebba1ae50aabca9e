#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c12; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_parity.py tests/test_hip_widen.py -m gpu -x -q -k "aten_sum_order or two_by_two or feather_strength_zero" 2>&1 | tail -15 | tee $O/pytest_new.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_all.log
