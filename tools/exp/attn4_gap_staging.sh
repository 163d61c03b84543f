# K / V staging of the 4-wave x 64-row attention body placed in its MFMA gaps (tools/exp/attn4_gap_staging.py has
# the what and why).  A/B in the patched scratch copy, same box, same process order: cycles per tile + clock from the in-kernel stamps
# (tools/exp/attn_stamps.sh), wall ms of the three bodies, and the body's parity tests on the gap-staged build.
# usage (GPU box): bash tools/exp/attn4_gap_staging.sh   -> gpurun_out/r4_attn4_gap_staging.log
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1
R=$GRAFT_REPO_ROOT
[ -f $R/.gapstage ] || { python3 $R/tools/exp/attn4_gap_staging.py $R/trajectorycrafter_amd/csrc/attn_fwd.hip && touch $R/.gapstage; } || exit 1
{
for f in "" "-DTCX_A4_GAPSTAGE" "" "-DTCX_A4_GAPSTAGE"; do
  echo "=== 4-wave body, staging: ${f:-super-step head / tail (round 3)}"
  bash $R/tools/exp/attn_stamps.sh "$f" 10 3 2>&1 | grep -v "^\[with_experiments\]"
done
# parity of the gap-staged build: the body's kernel tests (vs the oracle and vs the 32x32x16 body, bit-repeatable)
cd $R/trajectorycrafter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. -DTCX_A4_GAPSTAGE -x hip -c attn_fwd.hip -o /tmp/attn_gs.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_gs.so tcx_api.o /tmp/attn_gs.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
cd $R
echo "=== parity, gap-staged build"
TCX_LIB=/tmp/libtcx_gs.so timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -k "body_16x16x32" 2>&1 | tail -3
} 2>&1 | tee $R/gpurun_out/r4_attn4_gap_staging.log
