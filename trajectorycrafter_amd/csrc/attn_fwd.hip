// Fused attention forward for gfx950 (K1 self-attention D=64, K3 cross-attention D=128).
//
// Work decomposition: one 512-thread workgroup = 8 waves = 256 query rows of one (batch, head);
// each wave owns 32 query rows.  K/V tiles of 64 keys are staged global -> registers -> LDS
// (issue-early / write-late, double buffered, one barrier per tile).
//
// MFMA plan (v_mfma_f32_32x32x16_bf16), "swapped" so that a query row lives on ONE lane:
//   S^T[kv, q] = K[kv, :] . Q[q, :]      A = K rows (ds_read_b128 from an XOR-swizzled LDS image)
//                                         B = Q rows (registers for the whole kernel)
//   -> lane (q = lane & 31) holds 2 x 16 scores of its query row: the row max is 31 v_max + one
//      v_permlane32_swap, no LDS; the row sum stays a per-lane partial until the epilogue.
//   O^T[d, q] += V^T[d, kv] . P^T[kv, q]  B = the S^T accumulator converted to bf16 in place (an
//                                         accumulator tile is directly the next MFMA's B operand),
//                                         A = V^T fragments by ds_read_b64_tr_b16 (hardware
//                                         transpose) from a row-major, XOR-swizzled V image.
// Softmax in fp32 with exp2 and the scale folded into one FMA; the O rescale is skipped (exactly)
// whenever no lane of the wave saw its running max move.
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
};

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

template <int D, bool OUT_F32>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = D / 8;              // 16-B chunks per row
    constexpr int TILEB = 64 * D * 2;      // bytes of one K (or V) tile
    constexpr int NLD = 64 * CH / 512;     // 16-B loads per thread per tile and operand
    constexpr int KS = D / 16;             // k-steps of QK^T
    constexpr int DT = D / 32;             // 32-wide d tiles of O^T

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const uint32_t id = xcd_remap(blockIdx.x, p.nwg);
    const uint32_t bh = id / p.nqb, qb = id - bh * p.nqb;
    const int b = bh / p.H, hd = bh - b * p.H;
    const int q0 = qb * 256 + wave * 32;

    const uint16_t* kbase = p.k + (int64_t)b * p.ksb + (int64_t)hd * p.ksh;
    const uint16_t* vbase = p.v + (int64_t)b * p.vsb + (int64_t)hd * p.vsh;

    // ---- Q fragments: B operand, lane holds Q[q0 + r][16 ks + 8 h .. +8] ----
    bf16x8 qf[KS];
    {
        int qrow = q0 + r;
        if (qrow >= p.Sq) qrow = p.Sq - 1;
        const uint16_t* qp = p.q + (int64_t)b * p.qsb + (int64_t)qrow * p.qss + (int64_t)hd * p.qsh + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    // ---- staging: thread -> (row, chunk) of the 64-row tile ----
    int ld_row[NLD], ld_ch[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * 512;
        ld_row[i] = idx / CH;
        ld_ch[i] = idx % CH;
    }
    u32x4 kreg[NLD], vreg[NLD];
    const int ntiles = (p.Sk + 63) >> 6;

    auto stage_load = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int kv = tile * 64 + ld_row[i];
            if (kv < p.Sk) {
                kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)kv * p.kss + 8 * ld_ch[i]);
                vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (int64_t)kv * p.vss + 8 * ld_ch[i]);
            } else {
                kreg[i] = u32x4{0, 0, 0, 0};
                vreg[i] = u32x4{0, 0, 0, 0};   // zero V rows so that P = 0 never meets garbage
            }
        }
    };
    auto stage_write = [&](int buf) {
        char* kb = smem + buf * 2 * TILEB;
        char* vb = kb + TILEB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            *reinterpret_cast<u32x4*>(kb + k_off<D>(ld_row[i], ld_ch[i])) = kreg[i];
            *reinterpret_cast<u32x4*>(vb + v_chunk_off<D>(ld_row[i], ld_ch[i])) = vreg[i];
        }
    };

    // ---- per-lane LDS read offsets ----
    // K (A operand of QK^T): row 32 t + r, chunk 2 ks + h
    int koff[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) koff[t][ks] = k_off<D>(32 * t + r, 2 * ks + h);
    // V^T (A operand of PV) via ds_read_b64_tr_b16: lane 4 q4 + pp of a 16-lane group supplies the
    // address of row (kvb + q4), columns 4 pp .. 4 pp + 3 of the group's 16-column block.
    const int gl = lane & 15, q4 = gl >> 2, pp = gl & 3, g = (lane >> 4) & 1;
    int voff[DT];   // offset for kv block base 0 (+ 4 h), d tile dt; kv base adds a multiple of the row bytes
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
    float m = -INFINITY, l = 0.f;
    const float c = p.scale_log2;

    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        const char* kb = smem + buf * 2 * TILEB;
        const char* vb = kb + TILEB;
        const bool more = tile + 1 < ntiles;
        if (more) stage_load(tile + 1);

        // ---- S^T = K Q^T ----
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[t][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kb + koff[t][ks]);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[t], 0, 0, 0);
            }
        }
        // ---- mask the ragged tail (keys >= Sk) ----
        if (tile == ntiles - 1 && (p.Sk & 63)) {
            const int kv0 = tile * 64 + 4 * h;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kv = kv0 + 32 * t + (i & 3) + 8 * (i >> 2);
                    if (kv >= p.Sk) s[t][i] = -INFINITY;
                }
        }
        // ---- online softmax (row = lane & 31; the two half-waves hold disjoint keys of it) ----
        float mx = s[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
        mx *= c;
        {
            const uint32_t u = __float_as_uint(mx);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m, mx);
        if (!__all(m_new == m)) {   // wave-uniform; exact: alpha == 1 whenever skipped
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
            m = m_new;
        }
        bf16x8 pf[2][2];
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][i], c, -m));
                ls += e;
                pf[t][i >> 3][i & 7] = (__bf16)e;
            }
        l += ls;

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int rowb = (32 * t + 16 * s2) * (D * 2);
                    auto p0 = (__attribute__((address_space(3))) s16x4*)(vb + voff[dt] + rowb);
                    auto p1 = (__attribute__((address_space(3))) s16x4*)(vb + voff[dt] + rowb + 8 * (D * 2));
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 av = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), pf[t][s2], o[dt], 0, 0, 0);
                }
        }

        if (more) stage_write(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: combine the two half-wave partial sums, normalise, store O[q][d] ----
    {
        const uint32_t u = __float_as_uint(l);
        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l;
    const int qrow = q0 + r;
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

template <int D, bool F32>
int launch(const AttnParams& p, hipStream_t st) {
    constexpr int lds = 4 * 64 * D * 2;
    static bool attr_done = false;  // idempotent; racing threads set the same value
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<D, F32>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<D, F32>), dim3(p.nwg), dim3(512), lds, st, p);
    TCX_LAUNCH_RET();
}

}  // namespace

extern "C" int tcx_attn_fwd(const void* q, const void* k, const void* v, void* o,
                            int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t D,
                            int64_t qsb, int64_t qss, int64_t qsh, int64_t ksb, int64_t kss, int64_t ksh,
                            int64_t vsb, int64_t vss, int64_t vsh, int64_t osb, int64_t oss, int64_t osh,
                            float scale, int32_t out_dtype, void* stream) {
    TCX_CHECK(q && k && v && o, TCX_E_NULL, "tcx_attn_fwd: null pointer");
    TCX_CHECK(D == 64 || D == 128, TCX_E_SHAPE, "tcx_attn_fwd: head dim %d not in {64,128}", D);
    TCX_CHECK(B > 0 && H > 0 && Sq > 0 && Sk > 0, TCX_E_SHAPE, "tcx_attn_fwd: empty shape B=%d H=%d Sq=%d Sk=%d", B, H, Sq, Sk);
    TCX_CHECK(out_dtype == TCX_BF16 || out_dtype == TCX_F32, TCX_E_DTYPE, "tcx_attn_fwd: bad out_dtype %d", out_dtype);
    TCX_CHECK(tcx_aligned16(q) && tcx_aligned16(k) && tcx_aligned16(v) && tcx_aligned16(o), TCX_E_ALIGN,
              "tcx_attn_fwd: pointers must be 16-byte aligned");
    const int64_t st[12] = {qsb, qss, qsh, ksb, kss, ksh, vsb, vss, vsh, osb, oss, osh};
    for (int i = 0; i < 12; ++i)
        TCX_CHECK(st[i] % 8 == 0 && st[i] >= 0, TCX_E_ALIGN, "tcx_attn_fwd: stride %d (=%lld) must be a non-negative multiple of 8", i, (long long)st[i]);
    AttnParams p;
    p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.o = o;
    p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk;
    p.qsb = qsb; p.qss = qss; p.qsh = qsh; p.ksb = ksb; p.kss = kss; p.ksh = ksh;
    p.vsb = vsb; p.vss = vss; p.vsh = vsh; p.osb = osb; p.oss = oss; p.osh = osh;
    p.scale_log2 = scale * 1.4426950408889634f;
    p.nqb = (uint32_t)((Sq + 255) / 256);
    const uint64_t nwg = (uint64_t)p.nqb * B * H;
    TCX_CHECK(nwg < (1ull << 31), TCX_E_SHAPE, "tcx_attn_fwd: grid too large");
    p.nwg = (uint32_t)nwg;
    hipStream_t s = (hipStream_t)stream;
    if (D == 64) return out_dtype == TCX_F32 ? launch<64, true>(p, s) : launch<64, false>(p, s);
    return out_dtype == TCX_F32 ? launch<128, true>(p, s) : launch<128, false>(p, s);
}
