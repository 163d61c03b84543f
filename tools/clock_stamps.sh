# In-kernel clock of the two MFMA kernels that carry 91 % of a step: DIAGNOSTIC builds (patched COPIES of
# attn_fwd.hip / gemm.hip; the shipped kernels execute no stamp) record  s_memtime (shader cycles) and s_memrealtime (100 MHz) around
# the main loop and print them for a few workgroups; clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6),
# after >= 2 s of back-to-back launches on random data.  Usage on the GPU box:  bash tools/clock_stamps.sh <tag>
# Writes gpurun_out/<tag>_clock.json (+ the raw stamp lines in gpurun_out/<tag>_clock_raw.log).
. "$(dirname "${BASH_SOURCE[0]}")/exp/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
tag=${1:-r3}
R=$GRAFT_REPO_ROOT
cd $R/trajectorycrafter_amd/csrc
python3 - <<'PY'
g = open("gemm.hip").read()
g = g.replace("    for (int kt2 = 0; kt2 < KT; kt2 += 2) iteration(std::false_type{}, kt2);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");",
              "    const unsigned long long tC0 = __builtin_amdgcn_s_memtime(), tR0 = __builtin_amdgcn_s_memrealtime();\n    __builtin_amdgcn_s_waitcnt(0xC07F);\n"
              "    for (int kt2 = 0; kt2 < KT; kt2 += 2) iteration(std::false_type{}, kt2);\n"
              "    const unsigned long long tC1 = __builtin_amdgcn_s_memtime(), tR1 = __builtin_amdgcn_s_memrealtime();\n    __builtin_amdgcn_s_waitcnt(0xC07F);\n"
              "    if ((blockIdx.x % 499 == 7) && lane == 0 && (wid == 0 || wid == 7))\n"
              "        printf(\"GSTAMP epi %d wg %d wave %d iters %d cycles %llu real %llu\\n\", EPI, (int)blockIdx.x, wid, KT / 2, tC1 - tC0, tR1 - tR0);\n"
              "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");")
assert g.count("GSTAMP") == 1
open("/tmp/gemm_clk.hip", "w").write(g)
PY
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I."
/opt/rocm/bin/hipcc $F -DTCX_ATTN_STAMP -x hip -c attn_fwd.hip -o /tmp/attn_clk.o && /opt/rocm/bin/hipcc $F -x hip -c /tmp/gemm_clk.hip -o /tmp/gemm_clk.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_aclk.so tcx_api.o /tmp/attn_clk.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_gclk.so tcx_api.o attn_fwd.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o /tmp/gemm_clk.o || exit 1
cd $R
# ~300 launches of 7 ms = 2 s of attention; the stamps of the LAST launches are the settled ones (tail of the log)
TCX_LIB=/tmp/libtcx_aclk.so python3 tools/attn_body_bench.py 25 6 > gpurun_out/${tag}_clock_attn.log 2>&1 || { tail -5 gpurun_out/${tag}_clock_attn.log; exit 1; }
TCX_LIB=/tmp/libtcx_aclk.so python3 tools/microbench.py cross --iters 400 >> gpurun_out/${tag}_clock_attn.log 2>&1
TCX_LIB=/tmp/libtcx_gclk.so python3 tools/gemm_bench.py 150 > gpurun_out/${tag}_clock_gemm.log 2>&1 || { tail -5 gpurun_out/${tag}_clock_gemm.log; exit 1; }
python3 - "$tag" <<'PY'
import json, re, statistics, sys
tag = sys.argv[1]
out = {"method": "diagnostic builds (tools/clock_stamps.sh): s_memtime / s_memrealtime x 100 MHz around the main loop, lane 0 of waves 0 and 7 of every 997th "
                 "(attention) / 499th (GEMM) workgroup, median over the last half of the stamps of >= 2 s of back-to-back launches on random data",
       "kernels": {}}
def med(lines, key):
    by = {}
    for l in lines:
        m = re.search(r"cycles (\d+) real (\d+)", l)
        k = key(l)
        if m and k is not None and int(m.group(2)) > 0:
            by.setdefault(k, []).append((int(m.group(1)), int(m.group(2))))
    res = {}
    for k, v in by.items():
        v = v[len(v) // 2:]
        res[k] = {"stamps": len(v), "median_cycles": statistics.median(c for c, _ in v), "median_us": statistics.median(r for _, r in v) / 100.0,
                  "clock_GHz": round(statistics.median(c / r * 0.1 for c, r in v), 4)}
    return res
a = [l for l in open(f"gpurun_out/{tag}_clock_attn.log") if l.startswith("ASTAMP")]
out["kernels"]["attention main loop (bound-centred), by MFMA body / head dim"] = med(a, lambda l: "body %s, D=%s" % re.search(r"ASTAMP body (\d+) D (\d+)", l).groups())
g = [l for l in open(f"gpurun_out/{tag}_clock_gemm.log") if l.startswith("GSTAMP")]
out["kernels"]["gemm_kernel main loop, by epilogue / K iterations"] = med(g, lambda l: "epi %s, %s iterations" % re.search(r"epi (\d+) .* iters (\d+)", l).groups())
json.dump(out, open(f"gpurun_out/{tag}_clock.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
