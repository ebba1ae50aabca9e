# steady-state kernel breakdown of the headline workload in the split-bf16 depth mode (development; the committed cut comes from tools/make_profiles.sh)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/x3prof; mkdir -p $O; cd /tmp
rm -rf $O/t; rocprofv3 --kernel-trace -d $O/t -o p -- python $R/bench.py --depth-dtype f32x3 --steps 6 --warmup 4 --no-cpu-baseline --no-sub-records --no-profile > $O/t.log 2>&1
DB=$(find $O/t -name "*_results.db" | head -1)
python $R/tools/steady_state.py $DB 4 40 > $O/steady.txt; rm -rf $O/t
head -45 $O/steady.txt | cut -c1-150
