# A/B of compile-time variants of attn_fwd.hip on one box: each variant = extra -D flags; builds a private libtcx copy per variant and
# runs tools/attn_body_bench.py for them in alternation (two rounds).  usage: bash tools/exp/attn_variants.sh "-DFLAG_A" "-DFLAG_B" ...
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
R=$GRAFT_REPO_ROOT
cd $R/trajectorycrafter_amd/csrc
i=0
for flags in "" "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. $flags -x hip -c attn_fwd.hip -o /tmp/attn_v$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_v$i.so tcx_api.o /tmp/attn_v$i.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
  i=$((i+1))
done
cd $R
for round in 1 2; do
  j=0
  for flags in "" "$@"; do
    echo "== round $round variant $j: '${flags:-shipped}'"
    TCX_LIB=/tmp/libtcx_v$j.so python3 tools/attn_body_bench.py 20 3 | tail -1
    j=$((j+1))
  done
done
