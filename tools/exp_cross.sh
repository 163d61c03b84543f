# Timing-only ablations of the cross-attention (D = 128) loop (results wrong by construction; never shipped).
# usage on the GPU box: bash tools/exp_cross.sh
. "$(dirname "${BASH_SOURCE[0]}")/exp/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
BASE="-DTCX_EXP_NOLOAD -DTCX_EXP_NOWRITE -DTCX_EXP_NOBARRIER"
i=0
for extra in "" "-DTCX_EXP_NOLOAD -DTCX_EXP_NOWRITE" "$BASE" "$BASE -DTCX_EXP_NOEXP" "$BASE -DTCX_EXP_NOLDS" "$BASE -DTCX_EXP_NOEXP -DTCX_EXP_NOLDS" "-DTCX_EXP_NOEXP" ""; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -x hip -c attn_fwd.hip -o /tmp/attnx_$i.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_x$i.so tcx_api.o /tmp/attnx_$i.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o && \
  echo "== ${extra:-(shipped)}" && TCX_LIB=/tmp/libtcx_x$i.so python3 $GRAFT_REPO_ROOT/tools/microbench.py cross --iters 20 2>&1 | grep "bound-centred"
done
