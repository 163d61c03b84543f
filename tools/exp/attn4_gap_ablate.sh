# What is left in the gap-staged 4-wave body (tools/exp/attn4_gap_staging.sh): timing-only ablations (wrong results) on top of
# -DTCX_A4_GAPSTAGE — no barrier, no LDS fragment reads, no row-sum MFMAs — cycles per tile from the in-kernel stamps.
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1
R=$GRAFT_REPO_ROOT
[ -f $R/.gapstage ] || { python3 $R/tools/exp/attn4_gap_staging.py $R/trajectorycrafter_amd/csrc/attn_fwd.hip && touch $R/.gapstage; } || exit 1
for f in "-DTCX_A4_GAPSTAGE" "-DTCX_A4_GAPSTAGE -DTCX_A4_NOBAR" "-DTCX_A4_GAPSTAGE -DTCX_A4_NOLDS" "-DTCX_A4_GAPSTAGE -DTCX_A4_NOSUM" "-DTCX_A4_GAPSTAGE -DTCX_A4_NOBAR -DTCX_A4_NOLDS" "-DTCX_A4_GAPSTAGE -DTCX_A4_NOEXP"; do
  echo "=== $f"
  bash $R/tools/exp/attn_stamps.sh "$f" 6 2 2>&1 | grep "body 4 wave 0\|median:"
done 2>&1 | tee $R/gpurun_out/r4_attn4_gap_ablate.log
