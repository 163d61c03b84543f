# Timing experiments on the conv_mfma kernel (some variants are deliberately wrong: timing only, never shipped).
# Usage on the GPU box: bash tools/exp_conv.sh "<flags of variant 1>" "<flags of variant 2>" ...   ("" = shipped)
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
i=0
for extra in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -x hip -c conv_mfma.hip -o /tmp/convm_$i.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_c$i.so tcx_api.o attn_fwd.o norm.o elementwise.o conv.o groupnorm.o warp.o gemm.o /tmp/convm_$i.o && \
  echo "== variant ${extra:-(shipped)}" && TCX_LIB=/tmp/libtcx_c$i.so python3 $GRAFT_REPO_ROOT/tools/vae_bench.py shapes decode 2>&1 | grep -v amdgpu.ids
done
