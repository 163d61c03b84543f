"""Timing probes for the shipped 8-wave attention body's K / V staging (patched scratch copy only; wrong results):
  -DTCX_EXP_WAITONLY : the global loads stay, the kernel WAITS for them where it would write them to LDS, no ds_write
                       (separates 'load latency exposed at the write point' from 'cost of the LDS writes')
  -DTCX_EXP_2SETS    : (correct results) two staging register sets: tiles written at the end of super-step s were loaded at the start of
                       super-step s - 1, i.e. two super-steps of flight time instead of one; same ring, same slots, same barrier"""
import sys
p = sys.argv[1]
full = open(p).read()
a = full.index("__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnParams p)")
b = full.index("// ---- the bound-centred D = 64 loop on v_mfma_f32_16x16x32_bf16")
pre, s, post = full[:a], full[a:b], full[b:]
def rep(old, new, n=1):
    global s
    assert s.count(old) == n, (s.count(old), old[:90])
    s = s.replace(old, new)
rep("        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(kbuf0 + slot * TILEB + klds[i]) = kreg[j][i];\n",
    "#ifdef TCX_EXP_WAITONLY\n        for (int i = 0; i < NLD; ++i) { u32x4 tmp_ = kreg[j][i]; asm volatile(\"\" :: \"v\"(tmp_)); }\n#else\n"
    "        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(kbuf0 + slot * TILEB + klds[i]) = kreg[j][i];\n#endif\n")
rep("        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(vbuf0 + slot * TILEB + vlds[i]) = vreg[j][i];\n",
    "#ifdef TCX_EXP_WAITONLY\n        for (int i = 0; i < NLD; ++i) { u32x4 tmp_ = vreg[j][i]; asm volatile(\"\" :: \"v\"(tmp_)); }\n#else\n"
    "        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(vbuf0 + slot * TILEB + vlds[i]) = vreg[j][i];\n#endif\n")
open(p, "w").write(pre + s + post)
print("wait probe applied to", p)

# ---- -DTCX_EXP_2SETS (D = 64 / TPB = 2 only; correct results) ----
# Written against the staging order BEFORE "write at the top" (commit 7d1a0b1 and earlier: loads at a super-step's top, writes just
# before its barrier).  The shipped kernel has since adopted what this probe led to; on a newer tree only the wait-only probe applies.
full = open(p).read()
pre, s, post = full[:full.index("__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnParams p)")], None, None
a = full.index("__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnParams p)")
b = full.index("// ---- the bound-centred D = 64 loop on v_mfma_f32_16x16x32_bf16")
pre, s, post = full[:a], full[a:b], full[b:]
if "load_k(J0, t0 + TPB + 1);\n        load_v(J0, t0 + TPB);" not in s:
    print("2SETS probe skipped: this tree already stages at the top of the super-step (the probe recorded in "
          "profiles/r4_attn8_wait_probe.txt was run on commit 7d1a0b1)")
    sys.exit(0)
rep("    u32x4 kreg[TPB][NLD], vreg[TPB][NLD];\n",
    "    u32x4 kreg[TPB][NLD], vreg[TPB][NLD];\n#ifdef TCX_EXP_2SETS\n    u32x4 kreg2[TPB][NLD], vreg2[TPB][NLD];\n#endif\n")
old = s[s.index("        constexpr int PH = decltype(ph)::value;\n        load_k(J0, t0 + TPB + 1);\n        load_v(J0, t0 + TPB);\n"):s.index("        if constexpr (TPB == 2) {\n            // the second tile's K slot ((PH + 2) % R) is resident")]
new = '''        constexpr int PH = decltype(ph)::value;
#ifdef TCX_EXP_2SETS
        static_assert(TPB == 2 || D == 128, "2SETS probe is written for TPB == 2");
        // two staging register sets: this super-step loads the tiles that the NEXT super-step writes (set A when PH == 0, B otherwise)
        auto ld2 = [&](u32x4 (&kr)[TPB][NLD], u32x4 (&vr)[TPB][NLD]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                kr[0][i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[i] + (t0 + 2 * TPB + 1) * ktile_bytes, 0, 0);
                vr[0][i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[i] + (t0 + 2 * TPB) * vtile_bytes, 0, 0);
                kr[TPB - 1][i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[i] + (t0 + 2 * TPB + 2) * ktile_bytes, 0, 0);
                vr[TPB - 1][i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[i] + (t0 + 2 * TPB + 1) * vtile_bytes, 0, 0);
            }
        };
        if constexpr (TPB == 2) {
            if constexpr (PH == 0) ld2(kreg, vreg); else ld2(kreg2, vreg2);
        } else {
            load_k(J0, t0 + TPB + 1);
            load_v(J0, t0 + TPB);
        }
#else
''' + old[len("        constexpr int PH = decltype(ph)::value;\n"):] + "#endif\n"
s = s.replace(old, new)
old = s[s.index("#ifndef TCX_EXP_NOWRITE\n        write_k(J0, (PH + TPB + 1) % R);"):s.index("#ifndef TCX_EXP_NOBARRIER\n        __syncthreads();\n#endif\n    };")]
new = '''#ifdef TCX_EXP_2SETS
        auto wr2 = [&](u32x4 (&kr)[TPB][NLD], u32x4 (&vr)[TPB][NLD]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                *reinterpret_cast<u32x4*>(kbuf0 + ((PH + TPB + 1) % R) * TILEB + klds[i]) = kr[0][i];
                *reinterpret_cast<u32x4*>(vbuf0 + ((PH + TPB) % R) * TILEB + vlds[i]) = vr[0][i];
                *reinterpret_cast<u32x4*>(kbuf0 + ((PH + TPB + 2) % R) * TILEB + klds[i]) = kr[TPB - 1][i];
                *reinterpret_cast<u32x4*>(vbuf0 + ((PH + TPB + 1) % R) * TILEB + vlds[i]) = vr[TPB - 1][i];
            }
        };
        if constexpr (TPB == 2) {
            if constexpr (PH == 0) wr2(kreg2, vreg2); else wr2(kreg, vreg);       // the set the PREVIOUS super-step loaded
        } else {
            write_k(J0, (PH + TPB + 1) % R);
            write_v(J0, (PH + TPB) % R);
        }
#else
''' + old + "#endif\n"
s = s.replace(old, new)
# prologue: the first super-step (PH = 0) writes set B at its end
rep("    __syncthreads();                      // slot 0 of K is overwritten at the end of the first super-step\n",
    '''    __syncthreads();                      // slot 0 of K is overwritten at the end of the first super-step
#ifdef TCX_EXP_2SETS
    if constexpr (TPB == 2) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            kreg2[0][i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[i] + (TPB + 1) * ktile_bytes, 0, 0);
            vreg2[0][i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[i] + TPB * vtile_bytes, 0, 0);
            kreg2[TPB - 1][i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[i] + (TPB + 2) * ktile_bytes, 0, 0);
            vreg2[TPB - 1][i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[i] + (TPB + 1) * vtile_bytes, 0, 0);
        }
    }
#endif
''')
open(p, "w").write(pre + s + post)
print("2SETS probe applied to", p)
