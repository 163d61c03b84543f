# HBM traffic of the self-attention kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2; MI355X_MICROARCH.md "rocprofv3 PMC slots").
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/microbench.py attn --iters 3 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/microbench.py attn --iters 3 > $R/gpurun_out/pmc_write.log 2>&1
timeout -k 10 250 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_l2 -- python3 $R/tools/microbench.py attn --iters 3 > $R/gpurun_out/pmc_l2.log 2>&1
