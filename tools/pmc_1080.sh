export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc1080_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc1080_write -o p -- $CMD > /dev/null 2>&1
ls $R/gpurun_out/pmc1080_*/
