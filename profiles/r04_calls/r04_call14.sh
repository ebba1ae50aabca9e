# round-4 GPU call 14: end-to-end render_sbs_3d test; library-selection probes on the depth leg (TunableOp GEMM search, MIOpen find mode)
export TMPDIR=/tmp
O=gpurun_out/c14; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_widen.py -m gpu -x -q -k "end_to_end or batched_steps" 2>&1 | tail -5 | tee $O/pytest.log
timeout 900 python tools/probe_net_tune.py gemm $O/tunable_gemm.csv > $O/gemm.log 2>&1; tail -5 $O/gemm.log
ls -la $O
timeout 300 python tools/probe_net_tune.py use $O/tunable_gemm.csv > $O/use.log 2>&1; tail -3 $O/use.log
timeout 420 python tools/probe_net_tune.py find > $O/find.log 2>&1; tail -3 $O/find.log
