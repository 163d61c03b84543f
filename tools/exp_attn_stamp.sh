# In-kernel cycle stamps of the attention loop (s_memtime): loop total, LDS-write segment, barrier wait.
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
cp $GRAFT_REPO_ROOT/tools/exp/attn_stamp.hip.txt /tmp/attn_stamp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I. -x hip -c /tmp/attn_stamp.hip -o /tmp/attn_s.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_s.so tcx_api.o /tmp/attn_s.o norm.o elementwise.o conv.o groupnorm.o warp.o gemm.o && \
TCX_LIB=/tmp/libtcx_s.so python3 $GRAFT_REPO_ROOT/tools/microbench.py attn --iters 1 2>&1 | grep -v amdgpu.ids | grep "STAMP\|PP wg\|bound" | head -20
