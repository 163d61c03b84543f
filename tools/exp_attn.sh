# A/B builds of the attention kernel
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
i=0
for extra in "-DTCX_ATTN_SUM_MFMA=0" "-DTCX_ATTN_SUM_MFMA=1"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -x hip -c attn_fwd.hip -o /tmp/attn_$i.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_$i.so tcx_api.o /tmp/attn_$i.o norm.o elementwise.o conv.o groupnorm.o && \
  echo "== $extra" && TCX_LIB=/tmp/libtcx_$i.so python3 $GRAFT_REPO_ROOT/tools/microbench.py attn 2>&1 | grep -v amdgpu.ids | grep -v SDPA && (cd $GRAFT_REPO_ROOT && TCX_LIB=/tmp/libtcx_$i.so python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -k attn 2>&1 | tail -2)
done
