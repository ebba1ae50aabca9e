export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
CMD="python $R/bench.py --workload 4k-dibr --steps 4 --warmup 2 --batch 4 --clip 4 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d $R/gpurun_out/pmc_sq -o p -- $CMD > /dev/null 2>&1
ls $R/gpurun_out/pmc_*/
