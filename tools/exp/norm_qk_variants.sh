# A/B of the q/k LayerNorm + RoPE token kernel: shipped (q and k of a token in one wave, 2 waves per SIMD) vs one operand per wave
# (-DTCX_NORM_EXP_QK_SPLIT: grid.z picks q or k, 4 waves per SIMD); tools/exp/norm_qk_packed.patch is A/B'd the same way
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
R=$GRAFT_REPO_ROOT
cd $R/trajectorycrafter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. -DTCX_NORM_EXP_QK_SPLIT -x hip -c norm.hip -o /tmp/norm_v.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_qk_variant.so tcx_api.o attn_fwd.o /tmp/norm_v.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
cd $R
TCX_LIB=/tmp/libtcx_qk_variant.so python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "qk_layernorm" 2>&1 | tail -1
for r in 1 2 3; do
  echo "== shipped"; python3 tools/exp/copy_ceiling.py 2>/dev/null | tail -1
  echo "== variant"; TCX_LIB=/tmp/libtcx_qk_variant.so python3 tools/exp/copy_ceiling.py 2>/dev/null | tail -1
done
