"""VERDICT r3 item 6, the one lever DESIGN §3.1 had left open for the 4-wave x 64-row attention body: its K / V staging moved OUT of
the super-step's head / tail (8 buffer_loads up front, 8 ds_write_b128 + the barrier at the end — the 435 cycles per tile that one
wave per SIMD cannot hide) INTO the MFMA gaps: every slot's last gap carries only one v_cvt_pk (4.5 of its 32 cycles), so the 8
light gaps of the super-step's first tile take one buffer_load each and the 8 of its second tile one ds_write_b128 each
(-DTCX_A4_GAPSTAGE).  Same ring, same slots, same barrier: the written slots are not read by any wave during the super-step
(DESIGN §3.1), so only instruction placement changes and results are bit-identical.

This edits attn_fwd.hip of the PATCHED SCRATCH COPY (tools/exp/with_experiments.sh; the 4-wave body does not exist in the product
tree).  usage: tools/exp/attn4_gap_staging.sh (GPU box)."""
import sys
p = sys.argv[1]
full = open(p).read()
a4 = full.index("void attn_fwd4_kernel(const AttnParams p)")
a4e = full.index("// Adds the parts of the split tail workgroups")
pre, s, post = full[:a4], full[a4:a4e], full[a4e:]              # only the 4-wave body is edited
def rep(old, new, n=1):
    global s
    assert s.count(old) == n, (s.count(old), old[:90])
    s = s.replace(old, new)
# slot: a staging action in the light gap
rep("    auto slot = [&](auto has_next, auto stc, auto blkc, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], bf16x8 (&kf)[2], bf16x8 (&vp)[2])\n",
    "    auto slot = [&](auto has_next, auto stc, auto blkc, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], bf16x8 (&kf)[2], bf16x8 (&vp)[2], auto stg)\n")
rep("        mm(4); cv(6);\n", "        mm(4); cv(6); stg();\n")
# tile: pass slot index 0..7
rep("    auto tile = [&](auto has_next, const char* kb, const char* vb, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2]) __attribute__((always_inline)) {\n",
    "    auto tile = [&](auto has_next, const char* kb, const char* vb, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], auto stage) __attribute__((always_inline)) {\n")
import re
order = ["I0{}, I0{}", "I0{}, I1{}", "I1{}, I0{}", "I1{}, I1{}", "I2{}, I0{}", "I2{}, I1{}", "I3{}, I0{}", "I3{}, I1{}"]
for i, o in enumerate(order):
    pat = re.compile(r"(        slot\(has_next, " + re.escape(o) + r", cur, nxt, \w+, \w+)\);")
    assert len(pat.findall(s)) == 1, o
    s = pat.sub(r"\1, [&]() __attribute__((always_inline)) { stage(std::integral_constant<int, %d>{}); });" % i, s)
# one_tile / tail: thread the functor through
rep("    auto one_tile = [&](auto masked, auto slot_k, auto slot_v, int t, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2]) __attribute__((always_inline)) {\n"
    "        tile(std::true_type{}, kbuf0 + decltype(slot_k)::value * TILEB, vbuf0 + decltype(slot_v)::value * TILEB, cur, nxt);\n",
    "    auto no_stage = [](auto) __attribute__((always_inline)) {};\n"
    "    auto one_tile = [&](auto masked, auto slot_k, auto slot_v, int t, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], auto stage) __attribute__((always_inline)) {\n"
    "        tile(std::true_type{}, kbuf0 + decltype(slot_k)::value * TILEB, vbuf0 + decltype(slot_v)::value * TILEB, cur, nxt, stage);\n")
rep("            one_tile(std::true_type{}, std::integral_constant<int, (PH + 1) % R>{}, std::integral_constant<int, PH % R>{}, t0, sa, sb);\n"
    "            tile(std::false_type{}, kbuf0, vbuf0 + ((PH + 1) % R) * TILEB, sb, sa);\n",
    "            one_tile(std::true_type{}, std::integral_constant<int, (PH + 1) % R>{}, std::integral_constant<int, PH % R>{}, t0, sa, sb, no_stage);\n"
    "            tile(std::false_type{}, kbuf0, vbuf0 + ((PH + 1) % R) * TILEB, sb, sa, no_stage);\n")
rep("            tile(std::false_type{}, kbuf0, vbuf0 + (PH % R) * TILEB, sa, sb);\n", "            tile(std::false_type{}, kbuf0, vbuf0 + (PH % R) * TILEB, sa, sb, no_stage);\n")
# super_step: loads in the first tile's light gaps, writes in the second tile's
old = s[s.index("#if !defined(TCX_A4_NOSTAGE) && !defined(TCX_A4_NOLOADS)\n        load_k(J0, t0 + TPB + 1);"):s.index("#if !defined(TCX_A4_NOSTAGE) && !defined(TCX_A4_NOBAR)")]
new = '''#ifdef TCX_A4_GAPSTAGE
        // piece I of the super-step's 8: J = I >> 2 (register set), W = (I >> 1) & 1 (0 = K, 1 = V), N = I & 1 (16-byte piece)
        auto st_load = [&](auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value, J = I >> 2, W = (I >> 1) & 1, N = I & 1;
            if constexpr (W == 0) kreg[J][N] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[N] + (t0 + TPB + 1 + J) * ktile_bytes, 0, 0);
            else vreg[J][N] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[N] + (t0 + TPB + J) * vtile_bytes, 0, 0);
        };
        auto st_write = [&](auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value, J = I >> 2, W = (I >> 1) & 1, N = I & 1;
            if constexpr (W == 0) *reinterpret_cast<u32x4*>(kbuf0 + ((PH + TPB + 1 + J) % R) * TILEB + klds[N]) = kreg[J][N];
            else *reinterpret_cast<u32x4*>(vbuf0 + ((PH + TPB + J) % R) * TILEB + vlds[N]) = vreg[J][N];
        };
        one_tile(masked, std::integral_constant<int, (PH + 1) % R>{}, std::integral_constant<int, PH % R>{}, t0, sa, sb, st_load);
        one_tile(masked, std::integral_constant<int, (PH + 2) % R>{}, std::integral_constant<int, (PH + 1) % R>{}, t0 + 1, sb, sa, st_write);
#else
''' + old.replace("t0, sa, sb);", "t0, sa, sb, no_stage);").replace("t0 + 1, sb, sa);", "t0 + 1, sb, sa, no_stage);") + "#endif\n"
s = s.replace(old, new)
open(p, "w").write(pre + s + post)
print("attn4 gap staging applied to", p)
