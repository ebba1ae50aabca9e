# round-3 GPU call 4: full GPU tests (default geometry), parity subset under the alternative tile heights, A/B of E1_TH x W1_TH
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c4; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
VD3D_E1_TH=14 VD3D_W1_TH=16 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_edge_cases.py -m gpu -q > $O/pytest_alt.log 2>&1
cd /tmp
for e in 30 14; do for w in 32 16; do
  VD3D_E1_TH=$e VD3D_W1_TH=$w timeout 300 python $R/bench.py --workload 4k-dibr --steps 8 --warmup 2 --no-cpu-baseline --no-pixel-overlap > $O/seq_e${e}_w${w}.json 2> $O/seq_e${e}_w${w}.err
done; done
VD3D_E1_TH=14 timeout 300 python $R/bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov_e14.json 2>/dev/null
VD3D_E1_TH=30 timeout 300 python $R/bench.py --workload 4k-dibr --steps 13 --warmup 2 --no-cpu-baseline > $O/ov_e30.json 2>/dev/null
for e in 30 14; do
  rm -rf $O/p_sq$e; VD3D_E1_TH=$e timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/p_sq$e -o p -- python $R/bench.py --workload 4k-dibr --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find $O/p_sq$e -name "*_results.db") > $O/pmc_sq_e$e.md 2>&1; rm -rf $O/p_sq$e
done
ls -la $O
