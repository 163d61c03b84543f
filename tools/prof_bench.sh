# Round-end evidence run on the GPU box: GPU tests, the driver's bench command (plain), the same command under rocprofv3
# --kernel-trace --stats, and the HBM-side PMC traffic of the attention microbench (separate FETCH_SIZE / WRITE_SIZE passes).
# usage: bash tools/prof_bench.sh <tag>      (writes gpurun_out/<tag>_*)
tag=${1:-r2}
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q > gpurun_out/${tag}_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/${tag}_gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${tag}_bench_prof.json 2> $R/gpurun_out/${tag}_bench_prof.err || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 250 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/${tag}_pmc_attn_$ctr -- python3 $R/tools/microbench.py attn --iters 2 > $R/gpurun_out/${tag}_pmc_attn_$ctr.log 2>&1 || exit 1
done
echo done
