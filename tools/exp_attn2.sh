# A/B of attention build variants on one box.  Usage on the GPU box: bash tools/exp_attn2.sh
. "$(dirname "${BASH_SOURCE[0]}")/exp/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
i=0
for extra in "" "-DTCX_EXP_DIRECT_OSTORE" "" "-DTCX_EXP_DIRECT_OSTORE"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -x hip -c attn_fwd.hip -o /tmp/attn_$i.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_$i.so tcx_api.o /tmp/attn_$i.o norm.o elementwise.o conv.o groupnorm.o warp.o gemm.o && \
  echo "== variant ${extra:-(shipped)}" && TCX_LIB=/tmp/libtcx_$i.so python3 $GRAFT_REPO_ROOT/tools/microbench.py attn --iters 10 2>&1 | grep -v amdgpu.ids | grep -v SDPA
done
