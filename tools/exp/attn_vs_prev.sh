# A/B of the working-tree attn_fwd.hip against another version of the file (e.g. `git show HEAD:... > some/file` before the call — .git
# does not travel to the GPU box): alternating runs of tools/attn_body_bench.py on one box.  usage: bash tools/exp/attn_vs_prev.sh PREV_FILE
R=$GRAFT_REPO_ROOT
prev=$R/$1
cd $R/trajectorycrafter_amd/csrc
cp $prev /tmp/attn_prev.hip
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I."
/opt/rocm/bin/hipcc $F -x hip -c /tmp/attn_prev.hip -o /tmp/attn_prev.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_prev.so tcx_api.o /tmp/attn_prev.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
cd $R
for round in 1 2 3; do
  echo "== round $round: previous"; TCX_LIB=/tmp/libtcx_prev.so python3 tools/attn_body_bench.py 20 3 | tail -1
  echo "== round $round: working tree"; python3 tools/attn_body_bench.py 20 3 | tail -1
done
echo "== cross-attention: previous"; TCX_LIB=/tmp/libtcx_prev.so python3 tools/microbench.py cross --iters 30 | sed -n 3p
echo "== cross-attention: working tree"; python3 tools/microbench.py cross --iters 30 | sed -n 3p
echo "== cross-attention: previous"; TCX_LIB=/tmp/libtcx_prev.so python3 tools/microbench.py cross --iters 30 | sed -n 3p
echo "== cross-attention: working tree"; python3 tools/microbench.py cross --iters 30 | sed -n 3p
