# round-3 GPU call 8: after the LDS-attribute fix -- depth e2e tests, DIBR stage timings and a kernel trace with the torch-CPU arithmetic
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c8; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_depth_e2e.py tests/test_hip_parity.py -m gpu -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
cd /tmp
timeout 300 python $R/bench.py --workload 4k-dibr --steps 8 --warmup 2 --no-cpu-baseline --no-pixel-overlap > $O/seq_4k.json 2> $O/seq_4k.err
timeout 300 python $R/bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov_4k.json 2>/dev/null
timeout 300 python $R/bench.py --workload 1080p-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov_1080.json 2>/dev/null
rm -rf $O/kt; timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/bench.py --workload 4k-dibr --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 30 > $O/kernel_stats.md 2>&1 || true
rm -rf $O/kt
ls -la $O
