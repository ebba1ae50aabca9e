# PMC passes of the split-bf16 GEMM probe (one shape): bash tools/pmc_gemm.sh   (on the GPU box; writes gpurun_out/gemm_pmc.md)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
CMD="python $R/tools/probe_gemm_x3.py 39088 one"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/gemm_sq -o p -- $CMD > $R/gpurun_out/gemm_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/gemm_sq2 -o p -- $CMD > $R/gpurun_out/gemm_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/gemm_g -o p -- $CMD > $R/gpurun_out/gemm_g.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/gemm_f -o p -- $CMD > $R/gpurun_out/gemm_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/gemm_w -o p -- $CMD > $R/gpurun_out/gemm_w.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/gemm_sq/p_results.db $R/gpurun_out/gemm_sq2/p_results.db $R/gpurun_out/gemm_g/p_results.db $R/gpurun_out/gemm_f/p_results.db $R/gpurun_out/gemm_w/p_results.db > $R/gpurun_out/gemm_pmc.md 2>&1
find $R/gpurun_out -name "*.db" -size +20M -delete
cat $R/gpurun_out/gemm_pmc.md
