# Timing experiments on the GEMM kernel (some variants are deliberately unsafe: timing only, never shipped).
# Usage on the GPU box: bash tools/exp_gemm.sh
. "$(dirname "${BASH_SOURCE[0]}")/exp/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
i=0
for extra in "" "-DTCX_GEMM_EXP_DIRECT_RES" "" "-DTCX_GEMM_EXP_DIRECT_RES"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -x hip -c gemm.hip -o /tmp/gemm_$i.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_$i.so tcx_api.o attn_fwd.o norm.o elementwise.o conv.o groupnorm.o warp.o /tmp/gemm_$i.o && \
  echo "== variant ${extra:-(shipped)}" && TCX_LIB=/tmp/libtcx_$i.so python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py 20 2>&1 | grep -v amdgpu.ids
done
