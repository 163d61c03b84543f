# A/B of compile-time variants of norm.hip on one box (alternating runs of tools/microbench.py rows).  usage: bash tools/exp/norm_variants.sh "-DFLAG" ...
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
R=$GRAFT_REPO_ROOT
cd $R/trajectorycrafter_amd/csrc
i=0
for flags in "" "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. $flags -x hip -c norm.hip -o /tmp/norm_v$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_nv$i.so tcx_api.o attn_fwd.o /tmp/norm_v$i.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
  i=$((i+1))
done
cd $R
for round in 1 2 3; do
  j=0
  for flags in "" "$@"; do
    echo "== round $round variant $j: '${flags:-shipped}'"
    TCX_LIB=/tmp/libtcx_nv$j.so python3 tools/microbench.py rows --iters 50 2>/dev/null | sed -n 1p
    j=$((j+1))
  done
done
