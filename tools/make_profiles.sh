# Collects the rocprofv3 evidence that profiles/ keeps (run on the GPU box through gpurun; outputs under gpurun_out/profiles_rXX/):
#   1. kernel-trace --stats of the DIBR-only roofline runs (configs[2] at the CLI defaults, and -- round 5 -- the GUI-default configurations in which W1
#      runs without feathering), steady-state cuts of the headline (4K + DA-V2-Base float32) and of the configs[4] chain
#   2. three separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ counters) per DIBR-only workload -- never combined with trace domains other than
#      --kernel-trace (MI355X_MICROARCH.md HBM / rocprofv3 section)
# usage: VD3D_COMMIT=<git short hash> bash tools/make_profiles.sh r05
export TMPDIR=/tmp
TAG=${1:-r05}
R=$PWD; O=$R/gpurun_out/profiles_$TAG; mkdir -p $O
cd /tmp
run_stats() {  # name, cmd  -> kernel-stats table; keeps the database path in $DB
  rm -rf $O/t_$1; rocprofv3 --kernel-trace --stats -d $O/t_$1 -o p -- $2 > $O/t_$1.log 2>&1
  DB=$(find $O/t_$1 -name "*_results.db" | head -1)
  python $R/tools/rocpd_summary.py $DB 30 > $O/${TAG}_$1_kernel_stats.md
}
run_steady() {  # name, cmd, steps
  rm -rf $O/t_$1; rocprofv3 --kernel-trace -d $O/t_$1 -o p -- $2 > $O/t_$1.log 2>&1
  DB=$(find $O/t_$1 -name "*_results.db" | head -1)
  { echo "# steady-state kernel breakdown of \`$2\` (last $3 steps of a rocprofv3 --kernel-trace run; tools/steady_state.py)"; echo; echo '```';
    python $R/tools/steady_state.py $DB $3 60; echo '```'; } > $O/${TAG}_$1_steady.md; rm -rf $O/t_$1
}
SPECS=""
WLS="4k-dibr 4k-dibr-gui 1080p-gui-defaults"
[ -n "$ONLY_STEADY" ] && WLS=""   # ONLY_STEADY=1: just the two steady-state cuts
for WL in $WLS; do
  N=$(echo $WL | tr '-' '_')
  run_stats $N "python $R/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-pixel-overlap"
  TR=$DB
  CMD="python $R/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-profile"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
    n=$(echo $c | cut -d" " -f1)
    rm -rf $O/p_${N}_$n; rocprofv3 --kernel-trace --pmc $c -d $O/p_${N}_$n -o p -- $CMD > /dev/null 2>&1
  done
  DBS=$(find $O/p_${N}_FETCH_SIZE $O/p_${N}_WRITE_SIZE $O/p_${N}_SQ_WAVES -name "*_results.db" | tr '\n' ',')
  python $R/tools/pmc_summary.py $(echo $DBS | tr ',' ' ') > $O/${TAG}_pmc_${N}.md
  SPECS="$SPECS $WL=${DBS}$TR"
done
[ -n "$ONLY_STEADY" ] || VD3D_COMMIT=${VD3D_COMMIT:-unknown} python $R/tools/pmc_to_json.py $TAG $SPECS > $O/pmc_latest.json
[ -n "$SKIP_STEADY" ] || run_steady 4k_dav2b_f32 "python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-sub-records --no-profile" 4
[ -n "$SKIP_STEADY" ] || run_steady 4k_dav2b_f32x3 "python $R/bench.py --depth-dtype f32x3 --steps 6 --warmup 4 --no-cpu-baseline --no-sub-records --no-profile" 4
[ -n "$SKIP_STEADY" ] || run_steady 4k_dav2b_fp16x2 "python $R/bench.py --depth-dtype f32h2 --steps 6 --warmup 4 --no-cpu-baseline --no-sub-records --no-profile" 4
[ -n "$SKIP_STEADY" ] || run_steady 1080p_esrgan4k "python $R/bench.py --upscale-only" 3
rm -rf $O/p_* $O/t_*
ls -la $O
