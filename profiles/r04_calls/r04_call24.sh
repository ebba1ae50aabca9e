# round-4 GPU call 24: direct tests of the DPT neck / head glue kernels against torch
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_depthprep.py -m gpu -x -q 2>&1 | tail -12
