# round-3 GPU call 5: committed profiles (kernel stats, steady-state headline / esrgan chain, PMC passes) + the default driver-style bench run
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c5; mkdir -p $O
VD3D_COMMIT=dfbb685 bash tools/make_profiles.sh r03 > $O/profiles.log 2>&1
cd $R
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
ls -la $O $R/gpurun_out/profiles_r03
