# Timing-only ablation builds of the self-attention kernel (results are wrong by construction; never shipped).
# Each -DTCX_EXP_* removes one ingredient of the loop: global loads, LDS writes, the barrier, the exponentials,
# the LDS fragment reads, the PV MFMAs.  Usage on the GPU box:  bash tools/exp_attn.sh
. "$(dirname "${BASH_SOURCE[0]}")/exp/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
BASE="-DTCX_EXP_NOLOAD -DTCX_EXP_NOWRITE -DTCX_EXP_NOBARRIER"
i=0
for extra in "" "-DTCX_EXP_NOLOAD" "-DTCX_EXP_NOLOAD -DTCX_EXP_NOWRITE" "$BASE" "$BASE -DTCX_EXP_NOEXP" "$BASE -DTCX_EXP_NOLDS" "$BASE -DTCX_EXP_NOEXP -DTCX_EXP_NOLDS"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -x hip -c attn_fwd.hip -o /tmp/attn_$i.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_$i.so tcx_api.o /tmp/attn_$i.o norm.o elementwise.o conv.o groupnorm.o && \
  echo "== baseline ${extra:-(none)}" && TCX_LIB=/tmp/libtcx_$i.so python3 $GRAFT_REPO_ROOT/tools/microbench.py attn 2>&1 | grep -v amdgpu.ids | grep -v SDPA
done
