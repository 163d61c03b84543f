# A/B of the D = 128 (cross-attention) loop: K fragments of the next 16-key step requested into the same registers right behind the
# QK^T MFMAs (default build) vs read at the point of use (-DTCX_EXP_NOKPREF); parity tests on the default build first
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_product_gpu.py -m gpu -x -q -k "cross or attn or attention" 2>&1 | tail -1
cd trajectorycrafter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -DTCX_EXP_NOKPREF -x hip -c attn_fwd.hip -o /tmp/attn_nk.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_nokpref.so tcx_api.o /tmp/attn_nk.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  echo "== K prefetch"; python3 tools/microbench.py cross --iters 30 2>&1 | grep "bound-centred"
  echo "== at use"; TCX_LIB=/tmp/libtcx_nokpref.so python3 tools/microbench.py cross --iters 30 2>&1 | grep "bound-centred"
done
