# Upper bound of any barrier / ring restructuring of the SHIPPED 8-wave attention body: timing-only builds (wrong results) without
# its barrier, without its LDS writes, without its global loads — cycles per tile, clock and wall from the in-kernel stamps.
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1
R=$GRAFT_REPO_ROOT
for f in "" "-DTCX_EXP_NOBARRIER" "-DTCX_EXP_NOWRITE" "-DTCX_EXP_NOLOAD" "-DTCX_EXP_NOBARRIER -DTCX_EXP_NOWRITE -DTCX_EXP_NOLOAD" ""; do
  echo "=== ${f:-shipped}"
  bash $R/tools/exp/attn_stamps.sh "$f" 8 3 2>&1 | grep "body 32 wave\|median:"
done 2>&1 | tee $R/gpurun_out/r4_attn8_barrier_bound.log
