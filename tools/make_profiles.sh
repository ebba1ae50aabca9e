# Collects the rocprofv3 evidence that profiles/ keeps (run on the GPU box through gpurun; outputs under gpurun_out/profiles_rXX/):
#   1. kernel-trace --stats of the 4K DIBR-only roofline run (configs[2]), of the headline (4K + DA-V2-Base float32) and of the
#      configs[4] chain (1080p depth + DIBR + Real-ESRGAN x4)
#   2. three separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ counters) of the 4K DIBR-only run -- never combined with trace domains
#      other than --kernel-trace (MI355X_MICROARCH.md HBM / rocprofv3 section)
# usage: VD3D_COMMIT=<git short hash> bash tools/make_profiles.sh r04
export TMPDIR=/tmp
TAG=${1:-r04}
R=$PWD; O=$R/gpurun_out/profiles_$TAG; mkdir -p $O
cd /tmp
DIBR="python $R/bench.py --workload 4k-dibr --steps 6 --warmup 2 --no-cpu-baseline --no-profile"
HEAD="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sub-records --no-profile"
run_stats() {  # name, cmd
  rm -rf $O/t_$1; rocprofv3 --kernel-trace --stats -d $O/t_$1 -o p -- $2 > $O/t_$1.log 2>&1
  DB=$(find $O/t_$1 -name "*_results.db" | head -1)
  python $R/tools/rocpd_summary.py $DB 30 > $O/${TAG}_$1_kernel_stats.md; rm -rf $O/t_$1
}
# steady-state variant: the trace of a run whose first steps hold MIOpen's find pass (naive reference convolutions, ~90 % of the raw
# kernel time) is cut to the last N steps by tools/steady_state.py -- what the net's own kernels cost per step
run_steady() {  # name, cmd, steps
  rm -rf $O/t_$1; rocprofv3 --kernel-trace -d $O/t_$1 -o p -- $2 > $O/t_$1.log 2>&1
  DB=$(find $O/t_$1 -name "*_results.db" | head -1)
  { echo "# steady-state kernel breakdown of \`$2\` (last $3 steps of a rocprofv3 --kernel-trace run; tools/steady_state.py)"; echo; echo '```';
    python $R/tools/steady_state.py $DB $3 60; echo '```'; } > $O/${TAG}_$1_steady.md; rm -rf $O/t_$1
}
SEQ="python $R/bench.py --workload 4k-dibr --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap"
run_stats 4k_dibr "$SEQ"
run_steady 4k_dav2b_f32 "python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-sub-records --no-profile" 4
run_steady 1080p_esrgan4k "python $R/bench.py --upscale-only" 3
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  n=$(echo $c | cut -d" " -f1)
  rm -rf $O/p_$n; rocprofv3 --kernel-trace --pmc $c -d $O/p_$n -o p -- $DIBR > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $(find $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_WAVES -name "*_results.db") > $O/${TAG}_pmc_4k_dibr_raw.md
VD3D_COMMIT=${VD3D_COMMIT:-unknown} python $R/tools/pmc_to_json.py $TAG $(find $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_WAVES -name "*_results.db") > $O/pmc_latest.json
rm -rf $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_SQ_WAVES
ls -la $O
