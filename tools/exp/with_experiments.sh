# Source this at the top of an A/B / ablation script:   . "$(dirname "${BASH_SOURCE[0]}")/exp/with_experiments.sh"  (or ../exp/...)
#
# The product sources ship ONE attention body and no timing-only switches.  Everything that was measured against
# them — the v_mfma_f32_16x16x32_bf16 and 4-wave x 64-row attention bodies (with their C-ABI flags, ops arguments and tests), the
# row-sum-on-the-matrix-pipe forms, in-kernel clock stamps (TCX_ATTN_STAMP), the -DTCX_EXP_* / -DTCX_A4_* / -DTCX_GEMM_EXP_* /
# -DTCX_CONV_EXP_* / -DTCX_NORM_EXP_* ablation builds (several of which compute wrong results by design) — lives in
# tools/exp/attn_gemm_experiments.patch.  This helper makes a scratch COPY of the repository, applies the patch to the copy, builds
# its libtcx_hip.so and re-points GRAFT_REPO_ROOT at the copy, so that the scripts (which compile variant objects out of
# $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc and run tools/*.py from $GRAFT_REPO_ROOT) work unchanged; gpurun_out/ of the copy is a
# link to the real one.  The product tree is never modified.
_real=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
_exp=${TCX_EXP_ROOT:-/tmp/tcx_exp_tree}
if [ ! -f "$_exp/.patched" ]; then
  rm -rf "$_exp" && mkdir -p "$_exp" || return 1 2>/dev/null || exit 1
  (cd "$_real" && tar -c --exclude=./gpurun_out --exclude=./.git --exclude='*.o' --exclude='*.so' --exclude=__pycache__ .) | tar -x -C "$_exp"
  (cd "$_exp" && patch -p1 -s < tools/exp/attn_gemm_experiments.patch) || { echo "experiments patch does not apply to this tree" >&2; return 1 2>/dev/null || exit 1; }
  mkdir -p "$_real/gpurun_out" && ln -s "$_real/gpurun_out" "$_exp/gpurun_out"
  (cd "$_exp" && python3 -m trajectorycrafter_amd.build > "$_exp/build.log" 2>&1) || { tail -20 "$_exp/build.log" >&2; return 1 2>/dev/null || exit 1; }
  touch "$_exp/.patched"
fi
export GRAFT_REPO_ROOT="$_exp" TCX_EXP_REAL_ROOT="$_real"
cd "$_exp"
echo "[with_experiments] running in the patched copy $_exp (product tree: $_real)" >&2
