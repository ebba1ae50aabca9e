#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c16; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_fuzz.py tests/test_hip_parity.py -m gpu -x -q -k "render_loop_fuzz or other_entry_points" 2>&1 | tail -15 | tee $O/pytest.log
