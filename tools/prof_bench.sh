# Round-end evidence run on the GPU box, part 2 (tests + plain bench are their own gpurun call): the driver's bench command under
# rocprofv3 --kernel-trace --stats, one real 50-step clip vs the bench formula, the MFMA-shape microbenchmark, the in-kernel clocks.
# usage: bash tools/prof_bench.sh <tag>      (writes gpurun_out/<tag>_*)
tag=${1:-r3}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${tag}_bench_prof.json 2> $R/gpurun_out/${tag}_bench_prof.err || exit 1
echo "profiled bench done"; cat $R/gpurun_out/${tag}_bench_prof.json | cut -c1-300
cd $R
timeout -k 10 300 python3 tools/full_clip.py $tag > gpurun_out/${tag}_full_clip.log 2>&1 || { tail -5 gpurun_out/${tag}_full_clip.log; exit 1; }
tail -1 gpurun_out/${tag}_full_clip.log | cut -c1-600
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/exp/mfma_shape_group.hip -o /tmp/mfma_shape_group && timeout -k 10 120 /tmp/mfma_shape_group | tee gpurun_out/${tag}_mfma_shape_group.log
timeout -k 10 400 bash tools/clock_stamps.sh $tag > gpurun_out/${tag}_clock_stamps.out 2>&1; echo "clock rc=$?"; tail -40 gpurun_out/${tag}_clock_stamps.out
echo done
