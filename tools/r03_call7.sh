# round-3 GPU call 7: torch-CPU numerics on the device (pow / sigmoid / sqrt restated bit for bit) -- device-vs-oracle unit tests first,
# then the whole GPU suite with the all-zero reference bars, then the DIBR stage timings with the new arithmetic
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c7; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_torch_math.py -m gpu -q > $O/pytest_math.log 2>&1
tail -3 $O/pytest_math.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
cd /tmp
timeout 300 python $R/bench.py --workload 4k-dibr --steps 8 --warmup 2 --no-cpu-baseline --no-pixel-overlap > $O/seq_4k.json 2> $O/seq_4k.err
timeout 300 python $R/bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov_4k.json 2>/dev/null
timeout 300 python $R/bench.py --workload 1080p-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov_1080.json 2>/dev/null
ls -la $O
