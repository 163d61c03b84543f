# A/B of the row kernels (norm.hip) on one box.  usage: bash tools/exp_norm.sh "<src[:flags]>" ...   e.g. "norm.hip" "norm_prev_exp.hip"
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
i=0
for spec in "$@"; do
  i=$((i+1)); src=${spec%%:*}; extra=""; [ "$spec" != "$src" ] && extra=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -x hip -c $src -o /tmp/norm_$i.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_n$i.so tcx_api.o attn_fwd.o /tmp/norm_$i.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o && \
  echo "== variant $spec" && TCX_LIB=/tmp/libtcx_n$i.so python3 $GRAFT_REPO_ROOT/tools/microbench.py rows --iters 30 2>&1 | grep "layernorm\|qk "
done
