#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c17; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "gui_default_configuration" 2>&1 | tail -15 | tee $O/pytest.log
