# what each instruction class costs the single wave of attn_fwd4_kernel: timing-only builds (wrong results), cycles per tile from stamps
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
for f in "" "-DTCX_A4_NOLOADS" "-DTCX_A4_NOWRITES" "-DTCX_A4_NOBAR" "-DTCX_A4_NOLOADS -DTCX_A4_NOWRITES" "-DTCX_A4_NOSTAGE" "$@"; do
  echo "=== ${f:-baseline}"
  bash tools/exp/attn_stamps.sh "$f" 6 2 2>&1 | grep "body 4 wave 0"
done
