export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
CMD="python $R/tools/probe_w1.py"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d $R/gpurun_out/w1_sq -o p -- $CMD > $R/gpurun_out/w1_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR -d $R/gpurun_out/w1_sq2 -o p -- $CMD > $R/gpurun_out/w1_sq2.log 2>&1
