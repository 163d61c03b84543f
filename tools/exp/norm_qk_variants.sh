# A/B of the q/k LayerNorm + RoPE token kernel: packed-fp32 form (default build) vs the scalar form (-DTCX_NORM_EXP_SCALAR_QK)
R=$GRAFT_REPO_ROOT
cd $R/trajectorycrafter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. -DTCX_NORM_EXP_SCALAR_QK -x hip -c norm.hip -o /tmp/norm_sc.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_scalar_qk.so tcx_api.o attn_fwd.o /tmp/norm_sc.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
cd $R
for r in 1 2 3; do
  echo "== packed"; python3 tools/exp/copy_ceiling.py 2>/dev/null | tail -1
  echo "== scalar"; TCX_LIB=/tmp/libtcx_scalar_qk.so python3 tools/exp/copy_ceiling.py 2>/dev/null | tail -1
done
