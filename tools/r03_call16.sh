# round-3 GPU call 16 (final): full GPU suite, default bench, phase / residency stamps and the sequential kernel trace of the final kernels (807bd4f)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c16; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -2 $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json
VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 120 python tools/probe_phases.py > $O/phases.log 2>&1
cd /tmp
rm -rf $O/kt; timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/bench.py --workload 4k-dibr --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*_results.db" | head -1) 30 > $O/r03_4k_dibr_kernel_stats.md 2>&1; rm -rf $O/kt
ls $O
