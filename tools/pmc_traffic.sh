# HBM-side traffic per kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (no trace domains), for the
# attention microbench and the GEMM bench.  Summarise with tools/pmc_parse.py.
# usage (on the GPU box): bash tools/pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for what in "attn:tools/microbench.py attn --iters 2" "gemm:tools/gemm_bench.py 2"; do
  tag=${what%%:*}; cmd=${what#*:}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 250 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_${tag}_$ctr -- python3 $R/$cmd > $R/gpurun_out/pmc_${tag}_$ctr.log 2>&1 || exit 1
  done
done
