# rocprofv3 kernel stats of one decode + two encodes (tools/vae_bench.py) and the HBM-side PMC traffic of the conv shapes.
# Separate passes per counter (no trace domains with --pmc).  usage on the GPU box: bash tools/prof_vae.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_vae -- python3 $R/tools/vae_bench.py decode encode > $R/gpurun_out/r2_prof_vae.log 2>&1 || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_conv_$ctr -- python3 $R/tools/vae_bench.py shapes > $R/gpurun_out/pmc_conv_$ctr.log 2>&1 || exit 1
done
