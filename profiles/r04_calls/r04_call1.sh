# round-4 GPU call 1: full GPU suite on the batched chain + multi-stream pixel passes, then the launch-shape sweep at 4K
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/probe_step.py --clip 8 --check 0:16:1 1:16:1 2:16:1 2:16:2 2:16:4 2:8:2 2:4:2 3:16:2 4:16:2 1:16:2 > $O/probe.log 2>&1; cat $O/probe.log | tail -14
