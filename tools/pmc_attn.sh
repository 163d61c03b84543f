# PMC passes for the attention kernel (separate rocprofv3 runs per counter group; no trace domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_a -- python3 $R/tools/microbench.py attn --iters 3 > $R/gpurun_out/pmc_a.log 2>&1
timeout -k 10 250 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_b -- python3 $R/tools/microbench.py attn --iters 3 > $R/gpurun_out/pmc_b.log 2>&1
timeout -k 10 250 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $R/gpurun_out/pmc_c -- python3 $R/tools/microbench.py attn --iters 3 > $R/gpurun_out/pmc_c.log 2>&1
