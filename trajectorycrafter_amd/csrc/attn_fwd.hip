// Fused attention forward for gfx950 (K1 self-attention D=64, K3 cross-attention D=128).
//
// Work decomposition: one 512-thread workgroup = 8 waves = 256 query rows of one (batch, head);
// each wave owns 32 query rows.  K/V tiles of 64 keys are staged global -> registers -> LDS
// (issue-early / write-late, ring of LDS slots, one barrier per 128 keys at D=64 / per 64 keys at D=128).
//
// MFMA plan (v_mfma_f32_32x32x16_bf16), "swapped" so that a query row lives on ONE lane:
//   S^T[kv, q] = K[kv, :] . Q[q, :]      A = K rows (ds_read_b128 from an XOR-swizzled LDS image)
//                                         B = Q rows (registers for the whole kernel)
//   -> lane (q = lane & 31) holds 2 x 16 scores of its query row: the row max is 16 v_max3 + one
//      v_permlane32_swap, no LDS; the row sum stays a per-lane partial until the epilogue.
//   O^T[d, q] += V^T[d, kv] . P^T[kv, q]  B = the S^T accumulator converted to bf16 in place (an
//                                         accumulator tile is directly the next MFMA's B operand),
//                                         A = V^T fragments by ds_read_b64_tr_b16 (hardware
//                                         transpose) from a row-major, XOR-swizzled V image.
//
// The kernel is VALU-issue bound (rocprofv3: VALU active 66 % vs MFMA busy 39 % of the cycles in the
// first version), so the loop is built to minimise VALU instructions per score element:
//   * software pipeline — S(t+1) is issued in the same basic block as the exponentials of S(t);
//   * the tile loop is unrolled by two so every LDS address is base-register + immediate;
//   * K/V global loads are buffer loads: per-lane offset fixed + one scalar tile offset, rows beyond
//     Sk return zero from the hardware range check (no 64-bit address or predicate VALU);
//   * online softmax in fp32 with exp2 and the scale folded into one FMA; the O rescale is deferred
//     until some row's max has grown by more than 2^6 (P <= 64 in the meantime: bf16 rounding is
//     relative, fp32 accumulators have the head-room), which makes it rare even on random data;
//   * FAST path (D = 64, scores already in log2 units because the q/k-LayerNorm+RoPE kernel stores q
//     pre-multiplied by scale*log2(e)): the running max enters as the INITIAL ACCUMULATOR of the QK^T
//     MFMA chain (a persistent register block holding -m), so P = exp2(S') needs no FMA.
//     Per score element that leaves exp2 + 1 add + 1/2 cvt_pk + 1/2 max3 on the VALU.
//
// ONE loop body ships.  The bodies and timing-only ablations that were measured against it (v_mfma_f32_16x16x32_bf16 tiles, 4 waves
// x 64 rows with asm MFMAs, row sums on the matrix pipe, LDS-DMA staging, in-kernel clock stamps, -DTCX_EXP_NO* builds) are patches
// under tools/exp/ (attn_gemm_experiments.patch restores every one of them; DESIGN §3.1 has their numbers).
#include <stdlib.h>
#include <type_traits>
#include "tcx_common.h"

namespace {

struct AttnParams {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* v;
    void* o;
    int32_t B, H, Sq, Sk;
    int64_t qsb, qss, qsh, ksb, kss, ksh, vsb, vss, vsh, osb, oss, osh;
    float scale_log2;
    uint32_t nqb, nwg;
    const float* k_sqmax;   // [B*H] max_k |k|^2 (FAST path, optional): enables the bound-centred loop
    // Tail split (bound-centred kernel only): the last `nwg - n_full` workgroups (what would run as a partly filled last
    // round: nwg mod #CU) are each cut into `split` parts along the keys; a part writes its un-normalised O and row sum to
    // `ws` and attn_combine_kernel adds the parts.  With the bound-centred softmax every part uses the same exponent origin
    // M = |q| max|k|, so combining is a plain sum.  split <= 1: off.
    float* ws;
    size_t ws_bytes;
    uint32_t n_full, split;
    uint32_t proven;        // TCX_ATTN_BOUND_PROVEN: the caller guarantees M < 60 for every row -> no predicate, no complement launch
};
// workspace of the tail split: per item (tail workgroup x part): O [256][D] fp32, then l [256] fp32 per item, then one flag word
__host__ __device__ inline size_t attn_ws_o_floats(uint32_t items, int D) { return (size_t)items * 256 * D; }

constexpr float kDeferLog2 = 6.0f;   // rescale only when a row max grew by more than this (log2 units)

template <int D>
__device__ __forceinline__ int k_off(int row, int ch) {
    if constexpr (D == 64)  // two 128-B rows share one 256-B bank row; XOR the 16 chunks of the pair
        return ((row >> 1) << 8) + ((((((row & 1) << 3) | ch)) ^ ((row >> 1) & 15)) << 4);
    else                    // 256-B rows
        return (row << 8) + ((ch ^ (row & 15)) << 4);
}

// byte offset of the 16-byte chunk `ch` of V row `row`
template <int D>
__device__ __forceinline__ int v_chunk_off(int row, int ch) {
    if constexpr (D == 64)
        return row * 128 + ((((ch >> 1) ^ (row & 2))) << 5) + ((ch & 1) << 4);
    else
        return row * 256 + ((ch ^ ((row & 3) << 2)) << 4);
}

template <int D, bool OUT_F32, bool FAST, int NW, bool BOUND>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnParams p) {
    static_assert(!BOUND || FAST, "the bound-centred loop needs log2-domain scores");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = D / 8;              // 16-B chunks per row
    constexpr int TILEB = 64 * D * 2;      // bytes of one K (or V) tile
    constexpr int NLD = 64 * CH / (64 * NW);   // 16-B loads per thread per tile and operand
    constexpr int KS = D / 16;             // k-steps of QK^T
    constexpr int DT = D / 32;             // 32-wide d tiles of O^T

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    uint32_t pb = blockIdx.x;
    int part = -1;                                   // >= 0: this workgroup computes part `part` of a split tail workgroup
    uint32_t item = 0;
    if constexpr (BOUND && !OUT_F32) {
        if (p.split > 1 && pb >= p.n_full) {
            item = pb - p.n_full;
            part = (int)(item % p.split);
            pb = p.n_full + item / p.split;
        }
    }
    const uint32_t id = xcd_remap(pb, p.nwg);
    const uint32_t bh = id / p.nqb, qb = id - bh * p.nqb;
    const int b = bh / p.H, hd = bh - b * p.H;
    const int q0 = qb * (32 * NW) + wave * 32;

    // ---- K / V buffer descriptors: wave-uniform base, hardware range check at the end of row Sk-1 ----
    const uint16_t* kbase = p.k + (int64_t)b * p.ksb + (int64_t)hd * p.ksh;
    const uint16_t* vbase = p.v + (int64_t)b * p.vsb + (int64_t)hd * p.vsh;
    int Sk = p.Sk;                                   // keys this workgroup sweeps: all, or the tiles of its part
    if (part >= 0) {
        const int T = (p.Sk + 63) >> 6;
        const int t_lo = (int)((int64_t)part * T / p.split), t_hi = (int)((int64_t)(part + 1) * T / p.split);
        kbase += (int64_t)t_lo * 64 * p.kss;
        vbase += (int64_t)t_lo * 64 * p.vss;
        Sk = min(p.Sk, t_hi * 64) - t_lo * 64;
    }
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)((((int64_t)Sk - 1) * p.kss + D) * 2), 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)((((int64_t)Sk - 1) * p.vss + D) * 2), 0x00020000);
    const int ktile_bytes = (int)(64 * p.kss * 2), vtile_bytes = (int)(64 * p.vss * 2);

    // ---- Q fragments: B operand, lane holds Q[q0 + r][16 ks + 8 h .. +8] ----
    bf16x8 qf[KS];
    {
        int qrow = q0 + r;
        if (qrow >= p.Sq) qrow = p.Sq - 1;
        const uint16_t* qp = p.q + (int64_t)b * p.qsb + (int64_t)qrow * p.qss + (int64_t)hd * p.qsh + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    // ---- staging: thread -> (row, chunk) of the 64-row tile; per-lane byte offsets are loop invariant ----
    int kvoff[NLD], vvoff[NLD], klds[NLD], vlds[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * (64 * NW);
        const int row = idx / CH, ch = idx % CH;
        kvoff[i] = (int)(row * p.kss * 2) + ch * 16;
        vvoff[i] = (int)(row * p.vss * 2) + ch * 16;
        klds[i] = k_off<D>(row, ch);
        vlds[i] = v_chunk_off<D>(row, ch);
    }
    constexpr int TPB = (D == 64) ? 2 : 1;  // 64-key tiles per barrier / staging round (D = 128 has no registers for 2)
    constexpr int R = 2 * TPB;              // LDS ring slots per operand; tile t lives in slot t % R
    u32x4 kreg[TPB][NLD], vreg[TPB][NLD];
    const int ntiles = (Sk + 63) >> 6;

    auto load_k = [&](auto jc, int tile) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;   // rows >= Sk read as zero (range check); tiles beyond the end too
#pragma unroll
        for (int i = 0; i < NLD; ++i) kreg[j][i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[i] + tile * ktile_bytes, 0, 0);
    };
    auto load_v = [&](auto jc, int tile) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;   // zero V rows beyond Sk: P = 0 never meets garbage
#pragma unroll
        for (int i = 0; i < NLD; ++i) vreg[j][i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[i] + tile * vtile_bytes, 0, 0);
    };
    char* const kbuf0 = smem;
    char* const vbuf0 = smem + R * TILEB;
    auto write_k = [&](auto jc, int slot) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(kbuf0 + slot * TILEB + klds[i]) = kreg[j][i];
    };
    auto write_v = [&](auto jc, int slot) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(vbuf0 + slot * TILEB + vlds[i]) = vreg[j][i];
    };

    // ---- per-lane LDS read bases (everything else is a compile-time immediate) ----
    // K (A operand of QK^T): row 32 t + r, chunk 2 ks + h.  t adds 32 rows = a constant; the XOR term
    // only involves r and (2 ks + h), so one base per ks.
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = k_off<D>(r, 2 * ks + h);
    constexpr int K_T_STRIDE = 32 * D * 2;
    // V^T (A operand of PV) via ds_read_b64_tr_b16: lane 4 q4 + pp of a 16-lane group supplies the
    // address of row (kvb + q4), columns 4 pp .. 4 pp + 3 of the group's 16-column block.
    const int gl = lane & 15, q4 = gl >> 2, pp = gl & 3, g = (lane >> 4) & 1;
    int voff[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int dcol = 32 * dt + 16 * g + 4 * pp;
        voff[dt] = v_chunk_off<D>(4 * h + q4, dcol >> 3) + ((dcol & 7) << 1);
    }

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    float m = FAST ? 0.f : -INFINITY, l = 0.f;
    const float c = p.scale_log2;
    // FAST: -m replicated over an accumulator block (C-in of every S tile) and the MFMA row-sum accumulator
    f32x16 minit;
#pragma unroll
    for (int i = 0; i < 16; ++i) minit[i] = 0.f;
    bool first = true;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};   // per-lane partial row sums since the last rescale (VALU row-sum paths)
    // Bound-centred variant of the FAST path: with M = |q_row| * max_k |k| >= every score of the row
    // (Cauchy-Schwarz; max_k |k|^2 per (batch, head) comes from the q/k-LayerNorm+RoPE kernel), P = exp2(S - M)
    // is <= 1 for the whole sweep: no running max, no per-tile max3 chain, no rescale branch.  Floating-point
    // P keeps its relative precision at any scale; the only hazard is underflow of a whole row, impossible while
    // M - max_k S <= 2 M < 120, so the wave falls back to the exact tracking loop if any of its rows has M >= 60.
    // Two launches share the work when k_sqmax is given: the BOUND kernel takes the workgroups whose 256 rows all
    // have M < 60 and returns at once otherwise; the exact kernel (BOUND = false) makes the same test and takes
    // the complement.  The predicate is workgroup-uniform and evaluated before any other barrier.
    constexpr bool bounded = BOUND;
    if constexpr (FAST) {
        if (p.k_sqmax) {
            float qsq = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float qv = (float)qf[ks][j];
                    qsq = __builtin_fmaf(qv, qv, qsq);
                }
            const uint32_t u = __float_as_uint(qsq);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            qsq = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            const float M = sqrtf(qsq * p.k_sqmax[bh]) * 1.002f + 1e-3f;
            const bool safe = p.proven ? true : __syncthreads_and(M < 60.0f) != 0;
            if constexpr (BOUND) {
                if constexpr (!OUT_F32) {                // a split part tells the combine kernel whether it computed
                    if (part >= 0 && tid == 0)
                        reinterpret_cast<uint32_t*>(p.ws + attn_ws_o_floats(p.split * (p.nwg - p.n_full), D) + (size_t)p.split * (p.nwg - p.n_full) * 256)[item] = safe ? 1u : 0u;
                }
                if (!safe) return;
                m = M;
                first = false;
                if constexpr (D != 128) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) minit[i] = -M;
                }
            } else {
                if (safe) return;
            }
        }
    }

    // S^T tile of one 64-key block: 2 x (32 keys x 32 queries); `ks0..ks1` selects a slice of the k-steps
    // The initial accumulator of a QK^T chain (FAST: the persistent block holding -m, so that S' = K Q^T - m comes straight out of
    // the MFMA chain; otherwise zero — bound-centred D = 128: |S| <= M < 60, P = exp2(S) needs no centring at all) is the C operand
    // of the chain's FIRST MFMA, whose destination is the S tile: no per-tile copy of 32 registers (that copy was 32 v_mov per
    // wave-tile = 18 % of the issue cycles of an issue-bound loop).
    f32x16 czero;
#pragma unroll
    for (int i = 0; i < 16; ++i) czero[i] = 0.f;
    auto chain_c = [&](bool first_of_chain, const f32x16& acc) __attribute__((always_inline)) -> f32x16 {
        if (!first_of_chain) return acc;
        if constexpr (FAST && !(BOUND && D == 128)) return minit;
        else return czero;
    };
    auto qk_part = [&](const char* kb, f32x16 (&s)[2], int ks0, int ks1) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = ks0; ks < ks1; ++ks) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kb + koff[ks] + t * K_T_STRIDE);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], chain_c(ks == 0, s[t]), 0, 0, 0);
            }
        }
    };

    // running max of a freshly computed tile, then the (deferred, rare) rescale of O and l
    auto row_max_and_rescale = [&](f32x16 (&s)[2]) __attribute__((always_inline)) {
        float m0 = s[0][0], m1 = s[0][1], m2 = s[1][0], m3 = s[1][1];   // 4 independent v_max3 chains
#pragma unroll
        for (int i = 2; i < 16; i += 4) {
            m0 = fmaxf(fmaxf(m0, s[0][i]), s[0][i + 1]);
            m1 = fmaxf(fmaxf(m1, s[0][i + 2]), s[0][i + 3]);
            m2 = fmaxf(fmaxf(m2, s[1][i]), s[1][i + 1]);
            m3 = fmaxf(fmaxf(m3, s[1][i + 2]), s[1][i + 3]);
        }
        float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        if constexpr (!FAST) mx *= c;
        const uint32_t u = __float_as_uint(mx);
        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        if constexpr (FAST) {
            // mx is relative to the current m (S' = S - m).  Re-centre when a row grew by more than the
            // deferral threshold, and unconditionally on the first tile (m starts at 0).
            if (first || !__all(mx <= kDeferLog2)) {            // wave-uniform
                const float delta = first ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);   // first tile: o = l = 0, alpha irrelevant but finite
                m += delta;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[t][i] -= delta;    // this tile's scores were taken against the old m
                l = (l + (ls[0] + ls[1]) + (ls[2] + ls[3])) * alpha;
                ls[0] = ls[1] = ls[2] = ls[3] = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) minit[i] = -m;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
                first = false;
            }
        } else {
            if (!__all(mx <= m + kDeferLog2)) {   // wave-uniform.  m = -inf on the first tile -> always taken there
                const float m_new = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - m_new);   // exp2(-inf) = 0 on the first tile (o = l = 0)
                l = (l + (ls[0] + ls[1]) + (ls[2] + ls[3])) * alpha;     // fold the pending partial sums, all at the old max
                ls[0] = ls[1] = ls[2] = ls[3] = 0.f;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
                m = m_new;
            }
        }
    };
    auto mask_tail = [&](f32x16 (&s)[2]) {          // keys >= Sk of the last tile
        const int kv0 = (ntiles - 1) * 64 + 4 * h;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kv = kv0 + 32 * t + (i & 3) + 8 * (i >> 2);
                if (kv >= Sk) s[t][i] = -INFINITY;
            }
    };

    // One tile: P = exp2(.) of `cur` in four 16-key steps, each feeding its PV (and row-sum) MFMAs at
    // once, INTERLEAVED with one quarter of the next tile's QK^T MFMAs, so that every wave's instruction
    // stream alternates matrix and vector work (two such waves per SIMD then keep both pipes busy;
    // bunched MFMAs made the barrier-locked partner waves collide on one pipe at a time).  The freshly
    // finished next tile is max-checked at the end, under the tail of the PV MFMAs.
    constexpr int KPS = KS / 4;                      // k-steps of the next tile's QK^T per 16-key step
    // LDS -> register fragment reads of one 16-key step (issued one step ahead of their MFMAs)
    auto read_frags = [&](auto has_next, const char* kb, const char* vb, int st, bf16x8 (&kf)[KPS][2], bf16x8 (&vf)[DT]) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(has_next)::value;
        if constexpr (NEXT) {
#pragma unroll
            for (int j = 0; j < KPS; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    kf[j][t] = *reinterpret_cast<const bf16x8*>(kb + koff[st * KPS + j] + t * K_T_STRIDE);
                }
        }
        const int rowb = (32 * (st >> 1) + 16 * (st & 1)) * (D * 2);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            auto p0 = (__attribute__((address_space(3))) s16x4*)(vb + voff[dt] + rowb);
            auto p1 = (__attribute__((address_space(3))) s16x4*)(vb + voff[dt] + rowb + 8 * (D * 2));
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 av = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            vf[dt] = __builtin_bit_cast(bf16x8, av);
        }
    };
    auto tile_body = [&](auto has_next, const char* kb, const char* vb, f32x16 (&cur)[2], f32x16 (&nxt)[2]) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(has_next)::value;
        constexpr bool PREF = (D == 64);             // D = 128 has no registers to spare for a second fragment set
        bf16x8 kfa[KPS][2], kfb[KPS][2], vfa[DT], vfb[DT];
        if constexpr (PREF) read_frags(has_next, kb, vb, 0, kfa, vfa);
        auto one_step = [&](int st, bf16x8 (&kf)[KPS][2], bf16x8 (&vf)[DT], bf16x8 (&kfn)[KPS][2], bf16x8 (&vfn)[DT]) __attribute__((always_inline)) {
            const int t = st >> 1, s2 = st & 1;
            if constexpr (PREF) {
                if (st < 3) read_frags(has_next, kb, vb, st + 1, kfn, vfn);  // next step's operands, in flight under this step
            } else {
                read_frags(has_next, kb, vb, st, kf, vf);
            }
            if constexpr (NEXT) {
#pragma unroll
                for (int j = 0; j < KPS; ++j)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        nxt[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j][tt], qf[st * KPS + j], chain_c(st == 0 && j == 0, nxt[tt]), 0, 0, 0);
            }
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float e;
                if constexpr (FAST) {
                    e = __builtin_amdgcn_exp2f(cur[t][8 * s2 + j]);
                    ls[j & 3] += e;
                } else {
                    e = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[t][8 * s2 + j], c, -m));
                    ls[j & 3] += e;
                }
                pf[j] = (__bf16)e;
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt], pf, o[dt], 0, 0, 0);
            asm volatile("" : "+v"(ls[0]), "+v"(ls[1]), "+v"(ls[2]), "+v"(ls[3]));   // pin the partial sums here: without a use
            // inside the loop (bound-centred variant) the add chains get sunk to the loop end and 64 exponentials stay live
            __builtin_amdgcn_sched_barrier(0);       // keep the four steps in this order (no re-bunching)
        };
        if constexpr (PREF) {
            one_step(0, kfa, vfa, kfb, vfb);
            one_step(1, kfb, vfb, kfa, vfa);
            one_step(2, kfa, vfa, kfb, vfb);
            one_step(3, kfb, vfb, kfa, vfa);
        } else {
            one_step(0, kfa, vfa, kfa, vfa);
            one_step(1, kfa, vfa, kfa, vfa);
            one_step(2, kfa, vfa, kfa, vfa);
            one_step(3, kfa, vfa, kfa, vfa);
        }
    };

    // Fine-grained variant of the tile body for the bound-centred D = 64 loop: every MFMA is followed by exactly
    // {2 exp, 2 add, 1 cvt} (36.5 issue cycles per 32-cycle MFMA, MI355X_MICROARCH.md issue costs) instead of
    // {2 MFMA, 8 exp + 8 add + 4 cvt, 2 MFMA}: with the bunched order the two barrier-locked waves of a SIMD want the
    // matrix pipe at the same time and the vector issue at the same time, and their times ADD; with the alternating
    // order they slip one MFMA apart and mesh.  The PV product of a 16-key step is delayed by one step (pprev / vprev:
    // its P fragment and V fragments stay in registers) so that it can alternate with the next step's exponentials;
    // the last one is flushed after the loop.
    // Row sums: 32 v_add per tile.  The two matrix-pipe forms that were measured (one ones . P^T 32x32x16 MFMA per 16-key step: -6 %;
    // eight v_mfma_f32_4x4x4_16b_bf16 per tile: -0.5 %) live in tools/exp/attn_gemm_experiments.patch, DESIGN §3.1.
    constexpr bool FINE = BOUND && FAST && D == 64;
    bf16x8 pprev, vprev[DT];
#pragma unroll
    for (int j = 0; j < 8; ++j) pprev[j] = (__bf16)0.0f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vprev[dt] = pprev;
    // K fragments of the NEXT tile's first step are fetched during this tile's last step when that K tile is already
    // resident (first tile of a super-step): otherwise every tile starts by waiting one LDS round trip for them
    bf16x8 kfa[2], kfb[2];
    auto tile_body_fine = [&](auto has_next, auto prefetched, const char* kb, const char* kb_after, const char* vb, f32x16 (&cur)[2],
                              f32x16 (&nxt)[2], auto stg) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(has_next)::value, PRE = decltype(prefetched)::value;
        auto read_k_from = [&](const char* base, int st, bf16x8 (&kf)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 2; ++t) kf[t] = *reinterpret_cast<const bf16x8*>(base + koff[st] + t * K_T_STRIDE);
        };
        auto read_k = [&](int st, bf16x8 (&kf)[2]) __attribute__((always_inline)) { read_k_from(kb, st, kf); };
        auto read_v = [&](int st, int dt, bf16x8& vf) __attribute__((always_inline)) {
            const int rowb = (32 * (st >> 1) + 16 * (st & 1)) * (D * 2);
            auto p0 = (__attribute__((address_space(3))) s16x4*)(vb + voff[dt] + rowb);
            auto p1 = (__attribute__((address_space(3))) s16x4*)(vb + voff[dt] + rowb + 8 * (D * 2));
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
            vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        if constexpr (NEXT) {
            if constexpr (!PRE) read_k(0, kfa);
        }
        auto step = [&](int st, bf16x8 (&kf)[2], bf16x8 (&kfn)[2]) __attribute__((always_inline)) {
            const int t = st >> 1, s2 = st & 1;
            bf16x8 pf, vcur[DT];
            u32x4 pw;
            auto soft2 = [&](int j0) __attribute__((always_inline)) {
                const float e0 = __builtin_amdgcn_exp2f(cur[t][8 * s2 + j0]);
                const float e1 = __builtin_amdgcn_exp2f(cur[t][8 * s2 + j0 + 1]);
                ls[j0 & 3] += e0;
                ls[(j0 + 1) & 3] += e1;
                uint32_t w = pack_bf16(e0, e1);
                asm volatile("" : "+v"(w));              // convert here, inside this MFMA gap (the compiler sinks all four to the step's end)
                pw[j0 >> 1] = w;
            };
            if constexpr (NEXT) {
                if (st < 3) read_k(st + 1, kfn);
                else if (kb_after) read_k_from(kb_after, 0, kfn);      // step 3: kfn is kfa, what the next tile's step 0 uses
            }
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vprev[0], pprev, o[0], 0, 0, 0);
            soft2(0);
            __builtin_amdgcn_sched_barrier(0);
            read_v(st, 0, vcur[0]);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vprev[1], pprev, o[1], 0, 0, 0);
            soft2(2);
            __builtin_amdgcn_sched_barrier(0);
            read_v(st, 1, vcur[1]);
            if constexpr (NEXT) nxt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[st], chain_c(st == 0, nxt[0]), 0, 0, 0);
            soft2(4);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) nxt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[st], chain_c(st == 0, nxt[1]), 0, 0, 0);
            soft2(6);
            asm volatile("" : "+v"(ls[0]), "+v"(ls[1]), "+v"(ls[2]), "+v"(ls[3]));
            __builtin_amdgcn_sched_barrier(0);
            pf = __builtin_bit_cast(bf16x8, pw);

            pprev = pf;
            vprev[0] = vcur[0];
            vprev[1] = vcur[1];
        };
        step(0, kfa, kfb);
        stg(std::integral_constant<int, 0>{});
        step(1, kfb, kfa);
        stg(std::integral_constant<int, 1>{});
        step(2, kfa, kfb);
        stg(std::integral_constant<int, 2>{});
        step(3, kfb, kfa);
        stg(std::integral_constant<int, 3>{});
    };

    // Software pipeline over 64-key tiles; staging and synchronisation in rounds of TPB tiles ("super-steps").
    // Tile t lives in ring slot t % R of its operand.  At the start of the super-step of tiles t0 .. t0+TPB-1 the
    // ring holds K[t0+1 .. t0+TPB] and V[t0 .. t0+TPB-1] and the staging registers hold (or are about to receive)
    // K[t0+TPB+1 .. t0+2TPB] and V[t0+TPB .. t0+2TPB-1], requested one super-step ago.  The super-step FIRST writes
    // those registers into the slots whose tiles the previous super-step finished with (every wave is past the
    // barrier that ended it), re-issues the registers' loads two rounds ahead (K[t0+2TPB+1 ..], V[t0+2TPB ..]), runs
    // its tiles (tile t: S(t+1) from K[t+1] interleaved with softmax/PV of tile t from V[t]) and ends with the only
    // barrier, which publishes the tiles written during it for the next super-step.  Writing EARLY in the super-step
    // (round 4, D = 64; spread over its steps, see SPREAD below) instead of just before the barrier takes the ds_writes
    // and their completion wait out of the barrier's shadow.  D = 128 keeps the round-1 order — loads at the top, writes just before the barrier: its
    // staging registers would otherwise live across the barrier and the kernel (245 VGPRs) spills.  PH = t0 % R is a
    // template constant, so every LDS address is base register + immediate.
    constexpr bool WRITE_AT_TOP = D == 64;
    constexpr std::integral_constant<int, 0> J0{};
    constexpr std::integral_constant<int, TPB - 1> J1{};
    f32x16 sa[2], sb[2];
    // MASK: this tile may compute the scores of the LAST key tile (ragged Sk).  The steady-state loop is instantiated with
    // MASK = false: hipcc if-converts `if (last) mask_tail(nxt)` into 32 v_cndmask + the predicate arithmetic executed on EVERY
    // tile (18 % of the issue cycles of this issue-bound loop); only the <= 2 super-steps before the end carry the check.
    auto no_stg = [](auto) __attribute__((always_inline)) {};
    auto one_tile = [&](auto bnd, auto masked, auto slot_k, auto slot_v, int tile, f32x16 (&cur)[2], f32x16 (&nxt)[2], auto prefetched,
                        const char* kb_after, auto stg) __attribute__((always_inline)) {
        if constexpr (FINE) tile_body_fine(std::true_type{}, prefetched, kbuf0 + decltype(slot_k)::value * TILEB, kb_after,
                                           vbuf0 + decltype(slot_v)::value * TILEB, cur, nxt, stg);
        else tile_body(std::true_type{}, kbuf0 + decltype(slot_k)::value * TILEB, vbuf0 + decltype(slot_v)::value * TILEB, cur, nxt);
        if constexpr (decltype(masked)::value) {
            if (tile + 1 == ntiles - 1 && (Sk & 63)) mask_tail(nxt);
        }
        if constexpr (!decltype(bnd)::value) row_max_and_rescale(nxt);   // bound-centred loop: the reference max never moves
        else __builtin_amdgcn_sched_barrier(0);                         // keep tiles apart (register pressure)
    };
    auto super_step = [&](auto bnd, auto masked, auto ph, int t0) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph)::value;
        auto write_staged = [&]() __attribute__((always_inline)) {     // K[t0+TPB+1 ..], V[t0+TPB ..] into the slots of finished tiles
            write_k(J0, (PH + TPB + 1) % R);
            write_v(J0, (PH + TPB) % R);
            if constexpr (TPB == 2) {
                write_k(J1, (PH + TPB + 2) % R);
                write_v(J1, (PH + TPB + 1) % R);
            }
        };
        constexpr int AHEAD = WRITE_AT_TOP ? 2 * TPB : TPB;
        constexpr bool SPREAD = WRITE_AT_TOP && FINE && TPB == 2;
        // SPREAD (the fine D = 64 loop): the four staged pieces (K J0, V J0, K J1, V J1) do not move in one burst at the top but one at
        // a time, each written and its registers re-loaded behind steps 0 and 2 of the super-step's two tiles: a burst of 4 x 8
        // ds_write_b128 right behind the barrier holds the LDS port for 256 cycles while all eight waves want their first
        // fragments (stand-alone 6.88 -> 6.80 ms for all four behind the first tile's steps, 6.70 ms spread over both tiles; one
        // instruction behind every step, write and load apart, 6.73: DESIGN §3.1)
        auto piece = [&](auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I == 0) { write_k(J0, (PH + TPB + 1) % R); load_k(J0, t0 + AHEAD + 1); }
            else if constexpr (I == 1) { write_v(J0, (PH + TPB) % R); load_v(J0, t0 + AHEAD); }
            else if constexpr (I == 2) { write_k(J1, (PH + TPB + 2) % R); load_k(J1, t0 + AHEAD + 2); }
            else { write_v(J1, (PH + TPB + 1) % R); load_v(J1, t0 + AHEAD + 1); }
        };
        auto spread = [&](auto ic) __attribute__((always_inline)) {      // first tile: pieces 0, 1 behind steps 0, 2
            constexpr int I = decltype(ic)::value;
            if constexpr (I == 0) piece(std::integral_constant<int, 0>{});
            else if constexpr (I == 2) piece(std::integral_constant<int, 1>{});
        };
        auto spread_b = [&](auto ic) __attribute__((always_inline)) {    // second tile: pieces 2, 3 behind steps 0, 2
            constexpr int I = decltype(ic)::value;
            if constexpr (I == 0) piece(std::integral_constant<int, 2>{});
            else if constexpr (I == 2) piece(std::integral_constant<int, 3>{});
        };
        if constexpr (!SPREAD) {
            if constexpr (WRITE_AT_TOP) write_staged();           // loaded a super-step ago
            load_k(J0, t0 + AHEAD + 1);
            load_v(J0, t0 + AHEAD);
            if constexpr (TPB == 2) {
                load_k(J1, t0 + AHEAD + 2);
                load_v(J1, t0 + AHEAD + 1);
            }
        }
        if constexpr (TPB == 2) {
            // the second tile's K slot ((PH + 2) % R) is resident for the whole super-step: its first fragments are prefetched
            if constexpr (SPREAD)
                one_tile(bnd, masked, std::integral_constant<int, (PH + 1) % R>{}, std::integral_constant<int, PH % R>{}, t0, sa, sb, std::false_type{},
                         kbuf0 + ((PH + 2) % R) * TILEB, spread);
            else
                one_tile(bnd, masked, std::integral_constant<int, (PH + 1) % R>{}, std::integral_constant<int, PH % R>{}, t0, sa, sb, std::false_type{},
                         kbuf0 + ((PH + 2) % R) * TILEB, no_stg);
            if constexpr (SPREAD)
                one_tile(bnd, masked, std::integral_constant<int, (PH + 2) % R>{}, std::integral_constant<int, (PH + 1) % R>{}, t0 + 1, sb, sa, std::true_type{},
                         nullptr, spread_b);
            else
                one_tile(bnd, masked, std::integral_constant<int, (PH + 2) % R>{}, std::integral_constant<int, (PH + 1) % R>{}, t0 + 1, sb, sa, std::true_type{},
                         nullptr, no_stg);
        } else {
            if constexpr (PH == 0) one_tile(bnd, masked, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, t0, sa, sb, std::false_type{}, nullptr, no_stg);
            else one_tile(bnd, masked, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, t0, sb, sa, std::false_type{}, nullptr, no_stg);
        }
        if constexpr (!WRITE_AT_TOP) write_staged();          // loaded at this super-step's top
        __syncthreads();
    };
    // tiles left after the last full super-step: fewer than TPB steps with a successor (everything they read is
    // already resident: no staging, no barrier), then the peeled last tile (no successor)
    auto tail = [&](auto bnd, auto ph, int t0) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph)::value;
        const int rem = (ntiles - 1) - t0;                   // 0 .. TPB-1
        if constexpr (TPB == 2) {
            if (rem == 1) {
                one_tile(bnd, std::true_type{}, std::integral_constant<int, (PH + 1) % R>{}, std::integral_constant<int, PH % R>{}, t0, sa, sb, std::false_type{}, nullptr, no_stg);
                if constexpr (FINE) tile_body_fine(std::false_type{}, std::false_type{}, kbuf0, nullptr, vbuf0 + ((PH + 1) % R) * TILEB, sb, sa, no_stg);
                else tile_body(std::false_type{}, kbuf0, vbuf0 + ((PH + 1) % R) * TILEB, sb, sa);
            } else {
                if constexpr (FINE) tile_body_fine(std::false_type{}, std::false_type{}, kbuf0, nullptr, vbuf0 + (PH % R) * TILEB, sa, sb, no_stg);
                else tile_body(std::false_type{}, kbuf0, vbuf0 + (PH % R) * TILEB, sa, sb);
            }
        } else {
            (void)rem;
            if constexpr (PH == 0) tile_body(std::false_type{}, kbuf0, vbuf0, sa, sb);
            else tile_body(std::false_type{}, kbuf0, vbuf0 + TILEB, sb, sa);
        }
    };

    // prologue: K[0 .. TPB], V[0 .. TPB-1] into their slots, S(0)
    load_k(J0, 0);
    load_v(J0, 0);
    if constexpr (TPB == 2) {
        load_k(J1, 1);
        load_v(J1, 1);
    }
    write_k(J0, 0);
    write_v(J0, 0);
    if constexpr (TPB == 2) {
        write_k(J1, 1);
        write_v(J1, 1);
    }
    load_k(J0, TPB);
    write_k(J0, TPB % R);
    __syncthreads();
    qk_part(kbuf0, sa, 0, KS);
    if (ntiles == 1 && (Sk & 63)) mask_tail(sa);
    if constexpr (!bounded) row_max_and_rescale(sa);
    __syncthreads();                      // slot 0 of K is overwritten at the top of the first super-step
    if constexpr (WRITE_AT_TOP) {          // what the first super-step writes at its top
        load_k(J0, TPB + 1);
        load_v(J0, TPB);
        if constexpr (TPB == 2) {
            load_k(J1, TPB + 2);
            load_v(J1, TPB + 1);
        }
    }

    auto run = [&](auto bnd) __attribute__((always_inline)) {
        int t0 = 0;
        // steady state: a pair of super-steps computes the scores up to tile t0 + 2 TPB; while that is not the last tile: no mask
        for (; t0 + 2 * TPB < ntiles - 1; t0 += 2 * TPB) {
            super_step(bnd, std::false_type{}, std::integral_constant<int, 0>{}, t0);
            super_step(bnd, std::false_type{}, std::integral_constant<int, TPB>{}, t0 + TPB);
        }
        if (t0 + 2 * TPB <= ntiles - 1) {                    // the pair that reaches the last tile exactly
            super_step(bnd, std::true_type{}, std::integral_constant<int, 0>{}, t0);
            super_step(bnd, std::true_type{}, std::integral_constant<int, TPB>{}, t0 + TPB);
            t0 += 2 * TPB;
        }
        if (t0 + TPB <= ntiles - 1) {
            super_step(bnd, std::true_type{}, std::integral_constant<int, 0>{}, t0);
            tail(bnd, std::integral_constant<int, TPB>{}, t0 + TPB);
        } else {
            tail(bnd, std::integral_constant<int, 0>{}, t0);
        }
    };
    run(std::integral_constant<bool, BOUND>{});
    if constexpr (FINE) {                 // the delayed PV product of the very last 16-key step
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vprev[dt], pprev, o[dt], 0, 0, 0);
    }
    l += (ls[0] + ls[1]) + (ls[2] + ls[3]);

    // ---- epilogue: combine the two half-wave partial sums, normalise, store O[q][d] ----
    {
        const uint32_t u = __float_as_uint(l);
        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if constexpr (BOUND && !OUT_F32) {
        if (part >= 0) {                  // split part: un-normalised O (fp32) and the row sum go to the workspace
            const uint32_t items = p.split * (p.nwg - p.n_full);
            float* wo = p.ws + ((size_t)item * 256 + wave * 32 + r) * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4*>(wo + 32 * dt + 8 * i + 4 * h) = f32x4{o[dt][4 * i], o[dt][4 * i + 1], o[dt][4 * i + 2], o[dt][4 * i + 3]};
            if (h == 0) p.ws[attn_ws_o_floats(items, D) + (size_t)item * 256 + wave * 32 + r] = l;
            return;
        }
    }
    const float inv = 1.0f / l;
    const int qrow = q0 + r;
    if constexpr (!OUT_F32) {
        // bf16 output through the wave's own 32 x (2 D)-byte LDS tile (16-byte chunk c of row q at c ^ (q & 7)) so that it
        // leaves as row-wise 16-byte stores (full lines) instead of 4 DT scattered 8-byte stores per lane (store-issue bound)
        constexpr int RB = D * 2, CPR = D / 8, RPI = 64 / CPR;      // row bytes, chunks per row, rows per store instruction
        __syncthreads();                               // every wave is past its last K / V fragment read
        char* ot = smem + wave * (32 * RB);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d0 = 32 * dt + 8 * i + 4 * h;
                u32x2 w;
                w[0] = pack_bf16(o[dt][4 * i] * inv, o[dt][4 * i + 1] * inv);
                w[1] = pack_bf16(o[dt][4 * i + 2] * inv, o[dt][4 * i + 3] * inv);
                *reinterpret_cast<u32x2*>(ot + r * RB + (((d0 >> 3) ^ (r & 7)) << 4) + ((d0 & 4) << 1)) = w;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int rl = lane / CPR, ch = lane % CPR;
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
            const int row = i * RPI + rl;
            const u32x4 val = *reinterpret_cast<const u32x4*>(ot + row * RB + ((ch ^ (row & 7)) << 4));
            const int qr = q0 + row;
            if (qr < p.Sq)
                *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.osb + (int64_t)qr * p.oss + (int64_t)hd * p.osh + 8 * ch) = val;
        }
        return;
    }
    if (qrow < p.Sq) {
        const int64_t ooff = (int64_t)b * p.osb + (int64_t)qrow * p.oss + (int64_t)hd * p.osh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d0 = 32 * dt + 8 * i + 4 * h;
                const float a0 = o[dt][4 * i] * inv, a1 = o[dt][4 * i + 1] * inv;
                const float a2 = o[dt][4 * i + 2] * inv, a3 = o[dt][4 * i + 3] * inv;
                if constexpr (OUT_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.o) + ooff + d0) = f32x4{a0, a1, a2, a3};
                } else {
                    u32x2 w;
                    w[0] = pack_bf16(a0, a1);
                    w[1] = pack_bf16(a2, a3);
                    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.o) + ooff + d0) = w;
                }
            }
    }
}

// Adds the parts of the split tail workgroups: O = sum_p O_p / sum_p l_p (same exponent origin in every part) -> bf16.
// One block = 16 query rows x 16 threads (4 columns each, D = 64).
__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnParams p) {
    constexpr int D = 64;
    const uint32_t tail = p.nwg - p.n_full, items = tail * p.split;
    const uint32_t w = blockIdx.x >> 4, rb = blockIdx.x & 15;
    const uint32_t* flag = reinterpret_cast<const uint32_t*>(p.ws + attn_ws_o_floats(items, D) + (size_t)items * 256);
    if (flag[w * p.split] == 0) return;              // not bound-safe: the exact kernel computes this workgroup
    const uint32_t id = xcd_remap(p.n_full + w, p.nwg);
    const uint32_t bh = id / p.nqb, qb = id - bh * p.nqb;
    const int b = bh / p.H, hd = bh - b * p.H;
    const int row = rb * 16 + (threadIdx.x >> 4), c = (threadIdx.x & 15) * 4;
    const int q = qb * 256 + row;
    if (q >= p.Sq) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    for (uint32_t part = 0; part < p.split; ++part) {
        const size_t it = (size_t)w * p.split + part;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.ws + (it * 256 + row) * D + c);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
        l += p.ws[attn_ws_o_floats(items, D) + it * 256 + row];
    }
    const float inv = 1.0f / l;
    u32x2 o;
    o[0] = pack_bf16(acc[0] * inv, acc[1] * inv);
    o[1] = pack_bf16(acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.osb + (int64_t)q * p.oss + (int64_t)hd * p.osh + c) = o;
}

// Tail-split geometry of the bound-centred D = 64 bf16 launch: {tail workgroups, parts}; parts <= 1: no split.
struct AttnSplit { uint32_t tail, split; };
static AttnSplit attn_split_plan(int64_t nwg, int32_t Sk, uint32_t ncu) {
    AttnSplit s{0, 1};
    if (ncu == 0 || nwg < (int64_t)ncu) return s;    // fewer workgroups than CUs: nothing to balance
    const uint32_t tail = (uint32_t)(nwg % ncu);
    if (tail == 0 || tail * 2 > ncu) return s;       // the last round is at least half full
    uint32_t split = ncu / tail;
    const uint32_t ntiles = (uint32_t)((Sk + 63) >> 6);
    if (split > 8) split = 8;
    while (split > 1 && ntiles / split < 16) --split;   // a part should still amortise its prologue / epilogue
    s.tail = tail;
    s.split = split;
    return s;
}
static size_t attn_ws_bytes(AttnSplit s, int D) {
    const uint32_t items = s.tail * s.split;
    return s.split > 1 ? (attn_ws_o_floats(items, D) + (size_t)items * 256 + items) * 4 : 0;
}

template <int D, bool F32, bool FAST, int NW, bool BOUND>
int launch_one(AttnParams p, hipStream_t st) {
    constexpr int lds = 2 * (D == 64 ? 4 : 2) * 64 * D * 2;   // K ring + V ring, R slots each
    static TcxPerDeviceOnce lds_attr;   // per (kernel instantiation, device)
    const int arc = tcx_ensure_dynamic_lds(lds_attr, reinterpret_cast<const void*>(&attn_fwd_kernel<D, F32, FAST, NW, BOUND>), lds, "tcx_attn_fwd");
    if (arc != TCX_OK) return arc;
    p.nqb = (uint32_t)((p.Sq + 32 * NW - 1) / (32 * NW));
    p.nwg = p.nqb * (uint32_t)(p.B * p.H);
    uint32_t grid = p.nwg;
    float* ws = p.ws;
    p.ws = nullptr; p.n_full = p.nwg; p.split = 1;
    if constexpr (BOUND && !F32 && D == 64 && NW == 8) {
        const AttnSplit sp = attn_split_plan(p.nwg, p.Sk, tcx_cu_count());
        if (ws && sp.split > 1 && p.ws_bytes >= attn_ws_bytes(sp, D)) {
            p.ws = ws; p.split = sp.split; p.n_full = p.nwg - sp.tail;
            grid = p.n_full + sp.tail * sp.split;
        }
    }
    hipLaunchKernelGGL((attn_fwd_kernel<D, F32, FAST, NW, BOUND>), dim3(grid), dim3(64 * NW), lds, st, p);
    if constexpr (BOUND && !F32 && D == 64 && NW == 8) {
        if (p.split > 1) hipLaunchKernelGGL(attn_combine_kernel, dim3((p.nwg - p.n_full) * 16), dim3(256), 0, st, p);
    }
    TCX_LAUNCH_RET();
}

template <int D, bool F32, bool FAST>
int launch(const AttnParams& p, hipStream_t st) {
    if constexpr (FAST) {
        if (p.k_sqmax) {                                  // bound-centred kernel + exact kernel on the complement
            const int rc = launch_one<D, F32, true, 8, true>(p, st);
            if (rc != TCX_OK || p.proven) return rc;      // proven bound: every workgroup was bound-safe, the complement is empty
        }
    }
    return launch_one<D, F32, FAST, 8, false>(p, st);
}

}  // namespace

extern "C" int64_t tcx_attn_fwd_workspace_bytes(int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t D, int32_t flags,
                                                int32_t has_k_sqmax, int32_t out_dtype) {
    if (D != 64 || !(flags & TCX_ATTN_LOG2_SCORES) || !has_k_sqmax || out_dtype != TCX_BF16 || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return 0;
    const int64_t nwg = (int64_t)((Sq + 255) / 256) * B * H;
    return (int64_t)attn_ws_bytes(attn_split_plan(nwg, Sk, tcx_cu_count()), 64);
}

extern "C" int tcx_attn_fwd(const void* q, const void* k, const void* v, void* o,
                            int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t D,
                            int64_t qsb, int64_t qss, int64_t qsh, int64_t ksb, int64_t kss, int64_t ksh,
                            int64_t vsb, int64_t vss, int64_t vsh, int64_t osb, int64_t oss, int64_t osh,
                            float scale, int32_t flags, const float* k_sqmax, int32_t out_dtype, void* stream) {
    return tcx_attn_fwd_ws(q, k, v, o, B, H, Sq, Sk, D, qsb, qss, qsh, ksb, kss, ksh, vsb, vss, vsh, osb, oss, osh, scale, flags, k_sqmax,
                           out_dtype, nullptr, 0, stream);
}

extern "C" int tcx_attn_fwd_ws(const void* q, const void* k, const void* v, void* o,
                               int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t D,
                               int64_t qsb, int64_t qss, int64_t qsh, int64_t ksb, int64_t kss, int64_t ksh,
                               int64_t vsb, int64_t vss, int64_t vsh, int64_t osb, int64_t oss, int64_t osh,
                               float scale, int32_t flags, const float* k_sqmax, int32_t out_dtype,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    TCX_CHECK(workspace == nullptr || (tcx_aligned16(workspace) && workspace_bytes >= 0), TCX_E_ALIGN, "tcx_attn_fwd: workspace must be 16-byte aligned");
    TCX_CHECK(q && k && v && o, TCX_E_NULL, "tcx_attn_fwd: null pointer");
    TCX_CHECK((flags & ~(TCX_ATTN_LOG2_SCORES | TCX_ATTN_BOUND_PROVEN)) == 0, TCX_E_SHAPE, "tcx_attn_fwd: unknown flags 0x%x", flags);
    const bool log2s = (flags & TCX_ATTN_LOG2_SCORES) != 0;
    TCX_CHECK(!(flags & TCX_ATTN_BOUND_PROVEN) || (log2s && k_sqmax), TCX_E_SHAPE,
              "tcx_attn_fwd: TCX_ATTN_BOUND_PROVEN needs TCX_ATTN_LOG2_SCORES and k_sqmax");
    TCX_CHECK(!log2s || scale == 1.0f, TCX_E_SHAPE, "tcx_attn_fwd: TCX_ATTN_LOG2_SCORES requires scale == 1 (got %g)", scale);
    TCX_CHECK(D == 64 || D == 128, TCX_E_SHAPE, "tcx_attn_fwd: head dim %d not in {64,128}", D);
    TCX_CHECK(B > 0 && H > 0 && Sq > 0 && Sk > 0, TCX_E_SHAPE, "tcx_attn_fwd: empty shape B=%d H=%d Sq=%d Sk=%d", B, H, Sq, Sk);
    TCX_CHECK(out_dtype == TCX_BF16 || out_dtype == TCX_F32, TCX_E_DTYPE, "tcx_attn_fwd: bad out_dtype %d", out_dtype);
    TCX_CHECK(tcx_aligned16(q) && tcx_aligned16(k) && tcx_aligned16(v) && tcx_aligned16(o), TCX_E_ALIGN,
              "tcx_attn_fwd: pointers must be 16-byte aligned");
    const int64_t st[12] = {qsb, qss, qsh, ksb, kss, ksh, vsb, vss, vsh, osb, oss, osh};
    for (int i = 0; i < 12; ++i)
        TCX_CHECK(st[i] % 8 == 0 && st[i] >= 0, TCX_E_ALIGN, "tcx_attn_fwd: stride %d (=%lld) must be a non-negative multiple of 8", i, (long long)st[i]);
    TCX_CHECK(kss >= D && vss >= D, TCX_E_SHAPE, "tcx_attn_fwd: k/v row stride must be >= D");
    // K / V of one (batch, head) are addressed with 32-bit buffer offsets (two tiles of look-ahead included)
    TCX_CHECK(((int64_t)Sk + 448) * kss * 2 < (1ll << 31) && ((int64_t)Sk + 448) * vss * 2 < (1ll << 31), TCX_E_SHAPE,
              "tcx_attn_fwd: Sk * row stride exceeds the 2 GiB buffer-addressing range");
    AttnParams p;
    p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.o = o;
    p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk;
    p.qsb = qsb; p.qss = qss; p.qsh = qsh; p.ksb = ksb; p.kss = kss; p.ksh = ksh;
    p.vsb = vsb; p.vss = vss; p.vsh = vsh; p.osb = osb; p.oss = oss; p.osh = osh;
    p.scale_log2 = log2s ? 1.0f : scale * 1.4426950408889634f;
    p.k_sqmax = log2s ? k_sqmax : nullptr;
    TCX_CHECK((uint64_t)((Sq + 127) / 128) * B * H < (1ull << 31), TCX_E_SHAPE, "tcx_attn_fwd: grid too large");
    p.nqb = 0; p.nwg = 0;       // set per launch geometry
    p.ws = (float*)workspace; p.ws_bytes = workspace ? (size_t)workspace_bytes : 0; p.n_full = 0; p.split = 1;
    p.proven = (flags & TCX_ATTN_BOUND_PROVEN) ? 1u : 0u;
    hipStream_t s = (hipStream_t)stream;
    if (D == 64 && log2s) return out_dtype == TCX_F32 ? launch<64, true, true>(p, s) : launch<64, false, true>(p, s);
    if (D == 64) return out_dtype == TCX_F32 ? launch<64, true, false>(p, s) : launch<64, false, false>(p, s);
    if (D == 128 && log2s) return out_dtype == TCX_F32 ? launch<128, true, true>(p, s) : launch<128, false, true>(p, s);
    return out_dtype == TCX_F32 ? launch<128, true, false>(p, s) : launch<128, false, false>(p, s);
}
