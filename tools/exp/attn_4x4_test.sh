# parity + A/B of the 4x4x4 row-sum build of the fine loop
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
R=$GRAFT_REPO_ROOT
cd $R/trajectorycrafter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. -DTCX_ATTN_FINE_SUM_4X4 -x hip -c attn_fwd.hip -o /tmp/attn_4.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_4x4.so tcx_api.o /tmp/attn_4.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
cd $R
TCX_LIB=/tmp/libtcx_4x4.so python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_product_gpu.py -m gpu -x -q -k "attn or attention" 2>&1 | tail -2
for r in 1 2 3; do
  echo "== shipped"; python3 tools/attn_body_bench.py 20 3 | tail -1
  echo "== 4x4x4 row sum"; TCX_LIB=/tmp/libtcx_4x4.so python3 tools/attn_body_bench.py 20 3 | tail -1
done
