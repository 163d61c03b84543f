# In-kernel stamps (s_memtime) of the GEMM: counts per main-loop iteration (2 K-tiles = 128 MFMAs per wave), prologue
# and epilogue, for waves 0 and 7 of two workgroups.  Patches a COPY of gemm.hip.  Usage on the GPU box: bash tools/exp_gemm_stamp.sh
. "$(dirname "${BASH_SOURCE[0]}")/exp/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
python3 - <<'PY'
s = open("gemm.hip").read()
s = s.replace("    // ---- prologue: K-tile 0 complete, three half-tiles of K-tile 1 in flight ----", "    const long long tP0 = clock64();\n    // ---- prologue: K-tile 0 complete, three half-tiles of K-tile 1 in flight ----")
s = s.replace("    for (int kt2 = 0; kt2 < KT; kt2 += 2) iteration(std::false_type{}, kt2);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");",
              "    const long long tL0 = clock64();\n    for (int kt2 = 0; kt2 < KT; kt2 += 2) iteration(std::false_type{}, kt2);\n    const long long tL1 = clock64();\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");")
s = s.replace("    if (EPI == 2 && p.gate_v) store_rows(std::true_type{});\n    else store_rows(std::false_type{});\n}",
              "    if (EPI == 2 && p.gate_v) store_rows(std::true_type{});\n    else store_rows(std::false_type{});\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    const long long tE = clock64();\n    if ((blockIdx.x == 40 || blockIdx.x == 1500) && lane == 0 && (wid == 0 || wid == 7))\n        printf(\"GSTAMP epi %d wg %d wave %d: prologue %lld, loop %lld (%d iterations -> %lld per iteration), epilogue %lld\\\\n\", EPI, (int)blockIdx.x, wid,\n               tL0 - tP0, tL1 - tL0, KT / 2, (tL1 - tL0) / (KT / 2), tE - tL1);\n}")
assert s.count("GSTAMP") == 1 and s.count("tL0") >= 3
open("/tmp/gemm_stamp.hip", "w").write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. -x hip -c /tmp/gemm_stamp.hip -o /tmp/gemm_s.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_s.so tcx_api.o attn_fwd.o norm.o elementwise.o conv.o groupnorm.o warp.o /tmp/gemm_s.o && \
TCX_LIB=/tmp/libtcx_s.so python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py 1 2>&1 | grep "GSTAMP\|tcx_gemm" | sort | uniq -c | sort -rn | head -40
