# round-3 GPU call 13: full GPU suite, refreshed profiles (kernel stats, steady-state cuts, PMC) at commit 84aa9a7, default bench
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c13; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
VD3D_COMMIT=84aa9a7 timeout 900 bash tools/make_profiles.sh r03 > $O/make_profiles.log 2>&1
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
