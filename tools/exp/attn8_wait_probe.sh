# Where do the 19 % go that the shipped 8-wave attention body loses to its K / V staging (tools/exp/attn8_barrier_bound.sh: loads +
# LDS writes removed -> 1353 cycles per tile against 1676, LDS writes of never-loaded registers -> 1664)?  tools/exp/attn8_wait_probe.py:
#   -DTCX_EXP_WAITONLY  loads stay and are WAITED for at the write point, no ds_write (timing only)
#   -DTCX_EXP_2SETS     two staging register sets = two super-steps of flight time for every load (correct results; parity run below)
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1
R=$GRAFT_REPO_ROOT
[ -f $R/.waitprobe ] || { python3 $R/tools/exp/attn8_wait_probe.py $R/trajectorycrafter_amd/csrc/attn_fwd.hip && touch $R/.waitprobe; } || exit 1
{
grep -q TCX_EXP_2SETS $R/trajectorycrafter_amd/csrc/attn_fwd.hip && SETS="-DTCX_EXP_2SETS" || SETS=""      # the 2SETS probe only exists on trees before write-at-top
for f in "" "-DTCX_EXP_WAITONLY" $SETS "" $SETS; do
  echo "=== ${f:-shipped}"
  bash $R/tools/exp/attn_stamps.sh "$f" 8 3 2>&1 | grep "body 32 wave\|median:"
done
cd $R/trajectorycrafter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. -DTCX_EXP_2SETS -x hip -c attn_fwd.hip -o /tmp/attn_2s.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_2s.so tcx_api.o /tmp/attn_2s.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
cd $R
echo "=== parity, 2SETS build (attention kernel tests + the full-size product chain)"
TCX_LIB=/tmp/libtcx_2s.so timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_product_gpu.py -m gpu -q -k "attn or attention" 2>&1 | tail -3
} 2>&1 | tee $R/gpurun_out/r4_attn8_wait_probe.log
