# round-3 GPU call 14: ESRGAN head / tail kernels -- unit tests, whole up-scale suite, configs[4] chain, timing
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c14; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_upscale.py tests/test_hip_depth_e2e.py -m gpu -q -x > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 300 python tools/probe_conv.py > $O/probe.log 2>&1
tail -3 $O/probe.log
timeout 600 python bench.py --upscale-only > $O/upscale.json 2> $O/upscale.err
tail -c 1500 $O/upscale.json
