// HBM-bound row kernels of the DiT blocks (gfx950): K4 LayerNorm(+AdaLN modulate) and
// K2 per-head q/k LayerNorm + 3-D RoPE.  One 64-lane wave per row, 16-byte loads over the
// (T*H*W, C) token-major layout, statistics and arithmetic in fp32, one rounding at the store.
#include "tcx_common.h"

namespace {

struct LnParams {
    const uint16_t* x;
    uint16_t* y;
    int32_t B, rows, C;
    int64_t xsb, ysb;
    const uint16_t *gamma, *beta, *shift_v, *scale_v, *shift_t, *scale_t;
    int64_t msb;
    int32_t text_len;
    float eps;
    int32_t rows_per_wave;     // row-looping kernel only (set by the launcher)
};

// NCH = 16-byte chunks per lane (C <= NCH * 512)
template <int NCH>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const LnParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)p.B * p.rows) return;
    const int b = (int)(row / p.rows), rr = (int)(row - (int64_t)b * p.rows);
    const uint16_t* xr = p.x + (int64_t)b * p.xsb + (int64_t)rr * p.C;
    uint16_t* yr = p.y + (int64_t)b * p.ysb + (int64_t)rr * p.C;
    const int nchunk = p.C >> 3;

    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = j * 64 + lane;
        if (ch < nchunk) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(xr + 8 * ch);
            unpack8(raw, v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[j][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
        }
    }
    const float invC = 1.0f / (float)p.C;
    const float mean = wave_sum(sum) * invC;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = j * 64 + lane;
        if (ch < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[j][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) * invC + p.eps);

    const bool text = rr < p.text_len;
    const uint16_t* sh = text ? p.shift_t : p.shift_v;
    const uint16_t* sc = text ? p.scale_t : p.scale_v;
    if (sh) sh += (int64_t)b * p.msb;
    if (sc) sc += (int64_t)b * p.msb;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = j * 64 + lane;
        if (ch < nchunk) {
            float g[8], be[8], s1[8], s2[8], out[8];
            if (p.gamma) unpack8(*reinterpret_cast<const u32x4*>(p.gamma + 8 * ch), g);
            if (p.beta) unpack8(*reinterpret_cast<const u32x4*>(p.beta + 8 * ch), be);
            if (sc) unpack8(*reinterpret_cast<const u32x4*>(sc + 8 * ch), s1);
            if (sh) unpack8(*reinterpret_cast<const u32x4*>(sh + 8 * ch), s2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = (v[j][e] - mean) * rstd;
                if (p.gamma) t *= g[e];
                if (p.beta) t += be[e];
                if (sc) t *= (1.0f + s1[e]);
                if (sh) t += s2[e];
                out[e] = t;
            }
            *reinterpret_cast<u32x4*>(yr + 8 * ch) = pack8(out);
        }
    }
}

// Row-looping variant for large inputs: the per-row kernel above issues four parameter loads (gamma, beta, scale, shift)
// for every load of x, and its loads die with the row.  Here a wave keeps its slices of the four parameter vectors in
// registers (raw bf16), walks RPW rows of one segment (batch item x {text, video}: one modulation) and has the next row's
// loads in flight while it reduces and writes the current one.  Same arithmetic, same order.


// Round 3: the row-looping kernel with the modulation folded ONCE per workgroup into two fp32 vectors in LDS,
//     A = gamma (1 + scale),   Bc = beta (1 + scale) + shift        =>   y = ((x - mean) rstd) A + Bc,
// (identical to ((x - mean) rstd gamma + beta)(1 + scale) + shift up to fp32 rounding order; one rounding to bf16 at the store as
// before).  Why: with the four bf16 parameter vectors resident in registers the round-2 kernel sat at 219 VGPRs = 2 waves per SIMD
// AND spent ~20 VALU operations per element (4 unpacks + 5 arithmetic per element in the apply pass alone): at 8 waves per CU
// neither the 437 MB of HBM traffic nor the ~63 us of vector work per launch was hidden behind the other (3.9 TB/s).  Here a lane
// reads its chunk's A / Bc by two ds_read_b128 each (two planes per vector, 16 bytes per lane: conflict-free), the apply pass is
// 3 VALU per element, and the kernel needs < 100 VGPRs: 24 KiB of LDS per workgroup -> 6 workgroups = 24 waves per CU.
template <int NCH>
__global__ __launch_bounds__(256) void ln_modulate_lds_kernel(const LnParams p) {
    extern __shared__ __attribute__((aligned(16))) char ln_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y >> 1, vid = blockIdx.y & 1;
    const int seg0 = vid ? p.text_len : 0, seg1 = vid ? p.rows : p.text_len;
    const int rb = seg0 + (int)blockIdx.x * (4 * p.rows_per_wave);          // first row of this workgroup
    if (rb >= seg1) return;                                                  // workgroup-uniform: before any barrier
    const int nchunk = p.C >> 3;
    f32x4* const Alo = reinterpret_cast<f32x4*>(ln_lds);                     // planes of nchunk x 16 bytes: A[8 ch + 0..3], A[8 ch + 4..7],
    f32x4* const Ahi = Alo + nchunk;                                         // Bc likewise
    f32x4* const Blo = Ahi + nchunk;
    f32x4* const Bhi = Blo + nchunk;
    {
        const uint16_t* sh = vid ? p.shift_v : p.shift_t;
        const uint16_t* sc = vid ? p.scale_v : p.scale_t;
        if (sh) sh += (int64_t)b * p.msb;
        if (sc) sc += (int64_t)b * p.msb;
        for (int ch = threadIdx.x; ch < nchunk; ch += 256) {
            float g[8], be[8], s1[8], s2[8], A[8], Bc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { g[e] = 1.f; be[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
            if (p.gamma) unpack8(*reinterpret_cast<const u32x4*>(p.gamma + 8 * ch), g);
            if (p.beta) unpack8(*reinterpret_cast<const u32x4*>(p.beta + 8 * ch), be);
            if (sc) unpack8(*reinterpret_cast<const u32x4*>(sc + 8 * ch), s1);
            if (sh) unpack8(*reinterpret_cast<const u32x4*>(sh + 8 * ch), s2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float m = 1.0f + s1[e];
                A[e] = g[e] * m;
                Bc[e] = __builtin_fmaf(be[e], m, s2[e]);
            }
            Alo[ch] = f32x4{A[0], A[1], A[2], A[3]};
            Ahi[ch] = f32x4{A[4], A[5], A[6], A[7]};
            Blo[ch] = f32x4{Bc[0], Bc[1], Bc[2], Bc[3]};
            Bhi[ch] = f32x4{Bc[4], Bc[5], Bc[6], Bc[7]};
        }
    }
    __syncthreads();
    const int r0 = rb + wv;                                                  // this wave: r0, r0 + 4, ...
    if (r0 >= seg1) return;
    const uint16_t* xb = p.x + (int64_t)b * p.xsb;
    uint16_t* yb = p.y + (int64_t)b * p.ysb;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    u32x4 raw[NCH], nraw[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = j * 64 + lane;
        raw[j] = ch < nchunk ? *reinterpret_cast<const u32x4*>(xb + (int64_t)r0 * p.C + 8 * ch) : zero4;
    }
    const float invC = 1.0f / (float)p.C;
    for (int i = 0; i < p.rows_per_wave; ++i) {
        const int r = r0 + 4 * i;
        if (r >= seg1) break;
        const bool more = i + 1 < p.rows_per_wave && r + 4 < seg1;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = j * 64 + lane;
            nraw[j] = (more && ch < nchunk) ? *reinterpret_cast<const u32x4*>(xb + (int64_t)(r + 4) * p.C + 8 * ch) : zero4;
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float v[8];
            unpack8(raw[j], v);                          // chunks past the row are zero: they add nothing to the sum
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[e];
        }
        const float mean = wave_sum(sum) * invC;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (j * 64 + lane < nchunk) {
                float v[8];
                unpack8(raw[j], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[e] - mean;
                    sq = __builtin_fmaf(d, d, sq);
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) * invC + p.eps);
        uint16_t* yr = yb + (int64_t)r * p.C;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = j * 64 + lane;
            if (ch < nchunk) {
                const f32x4 a0 = Alo[ch], a1 = Ahi[ch], c0 = Blo[ch], c1 = Bhi[ch];
                float v[8], out[8];
                unpack8(raw[j], v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    out[e] = __builtin_fmaf((v[e] - mean) * rstd, a0[e], c0[e]);
                    out[4 + e] = __builtin_fmaf((v[4 + e] - mean) * rstd, a1[e], c1[e]);
                }
                *reinterpret_cast<u32x4*>(yr + 8 * ch) = pack8(out);
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) raw[j] = nraw[j];
    }
}

struct QkParams {
    uint16_t *q, *k;
    int32_t B, S, H;
    int64_t sb, ss, sh;
    const uint16_t *gq, *bq, *gk, *bk;
    const float *cos, *sin;
    int32_t text_len;
    float eps;
    int64_t nvec;
    float q_scale;
    float* k_sqmax;
    int32_t tokens_per_wave;   // token-per-wave kernel only (set by the launcher)
};

// 8 lanes per 64-wide head vector (8 bf16 = 16 B per lane); a wave covers 8 head vectors.
__global__ __launch_bounds__(256) void qk_ln_rope_kernel(const QkParams p) {
    const int lane = threadIdx.x & 63, sub = lane & 7;
    const int64_t vec = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const bool active = vec < p.nvec;
    // vec = ((b*S + s)*2 + which)*H + h
    const int64_t vv = active ? vec : 0;
    const int hh = (int)(vv % p.H);
    const int64_t t1 = vv / p.H;
    const int which = (int)(t1 & 1);
    const int64_t tok = t1 >> 1;
    const int b = (int)(tok / p.S), s = (int)(tok - (int64_t)b * p.S);
    uint16_t* base = (which ? p.k : p.q) + (int64_t)b * p.sb + (int64_t)s * p.ss + (int64_t)hh * p.sh + 8 * sub;
    float x[8];
    unpack8(*reinterpret_cast<const u32x4*>(base), x);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += x[e];
    sum = group8_sum(sum);
    const float mean = sum * (1.0f / 64.0f);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float d = x[e] - mean;
        sq += d * d;
    }
    sq = group8_sum(sq);
    const float rstd = rsqrtf(sq * (1.0f / 64.0f) + p.eps);
    float g[8], be[8];
    unpack8(*reinterpret_cast<const u32x4*>((which ? p.gk : p.gq) + 8 * sub), g);
    unpack8(*reinterpret_cast<const u32x4*>((which ? p.bk : p.bq) + 8 * sub), be);
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (x[e] - mean) * rstd * g[e] + be[e];
    if (p.cos && s >= p.text_len) {
        const int64_t ro = (int64_t)(s - p.text_len) * 64 + 8 * sub;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(p.cos + ro), c1 = *reinterpret_cast<const f32x4*>(p.cos + ro + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.sin + ro), s1 = *reinterpret_cast<const f32x4*>(p.sin + ro + 4);
        const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        float z[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // adjacent pairs (2i, 2i+1): rot = (-x_imag, x_real)
            z[2 * i] = y[2 * i] * cs[2 * i] - y[2 * i + 1] * sn[2 * i];
            z[2 * i + 1] = y[2 * i + 1] * cs[2 * i + 1] + y[2 * i] * sn[2 * i + 1];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = z[e];
    }
    if (!which) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] *= p.q_scale;
    }
    const u32x4 packed = pack8(y);
    if (active) *reinterpret_cast<u32x4*>(base) = packed;
    if (p.k_sqmax) {          // max_k |k|^2 of the ROUNDED keys per (batch, head): the attention kernel's Cauchy-Schwarz bound
        float r[8];
        unpack8(packed, r);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(r[e], r[e], ss);
        ss = group8_sum(ss);
        if (active && which && sub == 0) {
            unsigned* dst = reinterpret_cast<unsigned*>(p.k_sqmax + (int64_t)b * p.H + hh);
            const unsigned bits = __float_as_uint(ss);            // non-negative floats order like their bit patterns
            if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
        }
    }
}

// Token-per-wave variant for H = 8 * NI heads (the 5B model: NI = 6).  The kernel above reloads the token's cos / sin
// slice and the four LayerNorm parameter slices for every head vector: 112 B requested per 16-B payload.  Here lane
// (hs = lane >> 3, sub = lane & 7) owns elements 8 sub .. 8 sub + 7 of heads hs, hs + 8, ... of BOTH q and k of one token:
// cos / sin are loaded once per token, gamma / beta once per wave, the 2 * NI row chunks of the next token are in flight
// while the current one is normalised, and max |k|^2 is kept per head across the wave's tokens: one atomic per head
// per wave instead of one per head vector.  Same arithmetic, same order as the kernel above.

template <int NI>
__global__ __launch_bounds__(256) void qk_ln_rope_tok_kernel(const QkParams p) {
    const int lane = threadIdx.x & 63, sub = lane & 7, hs = lane >> 3;
    const int b = blockIdx.y;
    const int s0 = ((int)blockIdx.x * 4 + (threadIdx.x >> 6)) * p.tokens_per_wave;
    if (s0 >= p.S) return;
    const int s1 = min(s0 + p.tokens_per_wave, p.S);
    uint16_t* qb = p.q + (int64_t)b * p.sb + 8 * sub;
    uint16_t* kb = p.k + (int64_t)b * p.sb + 8 * sub;
    const u32x4 rgq = *reinterpret_cast<const u32x4*>(p.gq + 8 * sub), rbq = *reinterpret_cast<const u32x4*>(p.bq + 8 * sub);
    const u32x4 rgk = *reinterpret_cast<const u32x4*>(p.gk + 8 * sub), rbk = *reinterpret_cast<const u32x4*>(p.bk + 8 * sub);
    float kmax[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) kmax[i] = 0.f;
    u32x4 rq[NI], rk[NI], nq[NI], nk[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int64_t off = (int64_t)s0 * p.ss + (int64_t)(8 * i + hs) * p.sh;
        rq[i] = *reinterpret_cast<const u32x4*>(qb + off);
        rk[i] = *reinterpret_cast<const u32x4*>(kb + off);
    }
    for (int s = s0; s < s1; ++s) {
        const bool more = s + 1 < s1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t off = (int64_t)(more ? s + 1 : s) * p.ss + (int64_t)(8 * i + hs) * p.sh;
            nq[i] = *reinterpret_cast<const u32x4*>(qb + off);
            nk[i] = *reinterpret_cast<const u32x4*>(kb + off);
        }
        const bool rot = p.cos && s >= p.text_len;
        float cs[8], sn[8];
        if (rot) {
            const int64_t ro = (int64_t)(s - p.text_len) * 64 + 8 * sub;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(p.cos + ro), c1 = *reinterpret_cast<const f32x4*>(p.cos + ro + 4);
            const f32x4 z0 = *reinterpret_cast<const f32x4*>(p.sin + ro), z1 = *reinterpret_cast<const f32x4*>(p.sin + ro + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[e] = c0[e]; cs[4 + e] = c1[e]; sn[e] = z0[e]; sn[4 + e] = z1[e]; }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                float x[8], g[8], be[8], y[8];
                unpack8(which ? rk[i] : rq[i], x);
                unpack8(which ? rgk : rgq, g);
                unpack8(which ? rbk : rbq, be);
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += x[e];
                sum = group8_sum(sum);
                const float mean = sum * (1.0f / 64.0f);
                float sq = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = x[e] - mean;
                    sq += d * d;
                }
                sq = group8_sum(sq);
                const float rstd = rsqrtf(sq * (1.0f / 64.0f) + p.eps);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (x[e] - mean) * rstd * g[e] + be[e];
                if (rot) {
                    float z[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // adjacent pairs (2e, 2e+1): rot = (-x_imag, x_real)
                        z[2 * e] = y[2 * e] * cs[2 * e] - y[2 * e + 1] * sn[2 * e];
                        z[2 * e + 1] = y[2 * e + 1] * cs[2 * e + 1] + y[2 * e] * sn[2 * e + 1];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = z[e];
                }
                if (!which) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] *= p.q_scale;
                }
                const u32x4 packed = pack8(y);
                const int64_t off = (int64_t)s * p.ss + (int64_t)(8 * i + hs) * p.sh;
                *reinterpret_cast<u32x4*>((which ? kb : qb) + off) = packed;
                if (which && p.k_sqmax) {           // |k|^2 of the ROUNDED key
                    float r[8];
                    unpack8(packed, r);
                    float ss = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(r[e], r[e], ss);
                    ss = group8_sum(ss);
                    kmax[i] = fmaxf(kmax[i], ss);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) { rq[i] = nq[i]; rk[i] = nk[i]; }
    }
    if (p.k_sqmax && sub == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            unsigned* dst = reinterpret_cast<unsigned*>(p.k_sqmax + (int64_t)b * p.H + 8 * i + hs);
            const unsigned bits = __float_as_uint(kmax[i]);           // non-negative floats order like their bit patterns
            if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
        }
    }
}

}  // namespace

extern "C" int tcx_layernorm_modulate(const void* x, void* y, int32_t B, int32_t rows, int32_t C,
                                      int64_t xsb, int64_t ysb, const void* gamma, const void* beta,
                                      const void* shift_v, const void* scale_v, const void* shift_t,
                                      const void* scale_t, int64_t msb, int32_t text_len, float eps, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_layernorm_modulate: null x/y");
    TCX_CHECK(B > 0 && rows > 0 && C > 0 && C % 8 == 0 && C <= 8192, TCX_E_SHAPE,
              "tcx_layernorm_modulate: need C %% 8 == 0 and C <= 8192 (B=%d rows=%d C=%d)", B, rows, C);
    TCX_CHECK(xsb % 8 == 0 && ysb % 8 == 0 && msb % 8 == 0, TCX_E_ALIGN, "tcx_layernorm_modulate: strides must be multiples of 8");
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(y) && tcx_aligned16(gamma) && tcx_aligned16(beta) && tcx_aligned16(shift_v) &&
                  tcx_aligned16(scale_v) && tcx_aligned16(shift_t) && tcx_aligned16(scale_t),
              TCX_E_ALIGN, "tcx_layernorm_modulate: pointers must be 16-byte aligned");
    if (text_len > 0)
        TCX_CHECK((shift_t != nullptr) == (shift_v != nullptr) && (scale_t != nullptr) == (scale_v != nullptr), TCX_E_NULL,
                  "tcx_layernorm_modulate: text modulation must be given iff video modulation is");
    LnParams p{(const uint16_t*)x, (uint16_t*)y, B, rows, C, xsb, ysb, (const uint16_t*)gamma, (const uint16_t*)beta,
               (const uint16_t*)shift_v, (const uint16_t*)scale_v, (const uint16_t*)shift_t, (const uint16_t*)scale_t,
               msb, text_len, eps};
    const int64_t total = (int64_t)B * rows;
    dim3 grid((unsigned)((total + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int nch = (C / 8 + 63) / 64;
    if (total >= 4096 && nch <= 6 && B <= 32767) {       // large inputs: row-looping kernel, parameters in registers
        const int tl = (shift_t || scale_t) ? text_len : 0;   // no text modulation: one segment per batch item
        p.text_len = tl;
        // Rows per wave.  Round 3, LDS-parameter kernel (tools/exp/norm_variants.sh, alternating A/B on one box, product shape):
        // 1: 0.096-0.098 ms | 2: 0.095 | 3: 0.097 | 4: 0.098 | 8: 0.106 | 16: 0.114 | 35 (one round): 0.117 — a wave has one row
        // in flight besides the one it reduces, so short waves (many workgroups, staggered phases) stream better, and the 24 KiB
        // parameter set-up per workgroup is cheap enough to repeat every 8 rows.  (The round-2 register-parameter kernel at 8 rows:
        // 0.112 ms; sizing IT to one resident round had measured 11 % slower.)
        const int seg = tl > rows - tl ? tl : rows - tl;
        const int64_t rpw = 2;
        p.rows_per_wave = (int32_t)rpw;
        const int per_block = 4 * (int)rpw;
        dim3 g2((unsigned)((seg + per_block - 1) / per_block), (unsigned)(2 * B));
        {
            const size_t lds = (size_t)C * 8;            // A and Bc as fp32: 24 KiB at C = 3072 (< the 64 KiB default limit up to C = 8192)
            if (nch <= 2) hipLaunchKernelGGL(ln_modulate_lds_kernel<2>, g2, block, lds, st, p);
            else if (nch <= 4) hipLaunchKernelGGL(ln_modulate_lds_kernel<4>, g2, block, lds, st, p);
            else hipLaunchKernelGGL(ln_modulate_lds_kernel<6>, g2, block, lds, st, p);
            TCX_LAUNCH_RET();
        }
    }
    if (nch <= 1) hipLaunchKernelGGL(ln_modulate_kernel<1>, grid, block, 0, st, p);
    else if (nch <= 2) hipLaunchKernelGGL(ln_modulate_kernel<2>, grid, block, 0, st, p);
    else if (nch <= 4) hipLaunchKernelGGL(ln_modulate_kernel<4>, grid, block, 0, st, p);
    else if (nch <= 6) hipLaunchKernelGGL(ln_modulate_kernel<6>, grid, block, 0, st, p);
    else if (nch <= 8) hipLaunchKernelGGL(ln_modulate_kernel<8>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(ln_modulate_kernel<16>, grid, block, 0, st, p);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_qk_layernorm_rope(void* q, void* k, int32_t B, int32_t S, int32_t H, int32_t D,
                                     int64_t sb, int64_t ss, int64_t sh,
                                     const void* gq, const void* bq, const void* gk, const void* bk,
                                     const float* cos, const float* sin, int32_t text_len, float eps, float q_scale,
                                     float* k_sqmax, void* stream) {
    TCX_CHECK(q && k && gq && bq && gk && bk, TCX_E_NULL, "tcx_qk_layernorm_rope: null pointer");
    TCX_CHECK(D == 64, TCX_E_SHAPE, "tcx_qk_layernorm_rope: head dim must be 64 (got %d)", D);
    TCX_CHECK(B > 0 && S > 0 && H > 0 && text_len >= 0 && text_len <= S, TCX_E_SHAPE, "tcx_qk_layernorm_rope: bad shape");
    TCX_CHECK((cos == nullptr) == (sin == nullptr), TCX_E_NULL, "tcx_qk_layernorm_rope: cos and sin must both be given or both null");
    TCX_CHECK(sb % 8 == 0 && ss % 8 == 0 && sh % 8 == 0, TCX_E_ALIGN, "tcx_qk_layernorm_rope: strides must be multiples of 8");
    TCX_CHECK(tcx_aligned16(q) && tcx_aligned16(k) && tcx_aligned16(gq) && tcx_aligned16(bq) && tcx_aligned16(gk) &&
                  tcx_aligned16(bk) && tcx_aligned16(cos) && tcx_aligned16(sin),
              TCX_E_ALIGN, "tcx_qk_layernorm_rope: pointers must be 16-byte aligned");
    QkParams p{(uint16_t*)q, (uint16_t*)k, B, S, H, sb, ss, sh, (const uint16_t*)gq, (const uint16_t*)bq,
               (const uint16_t*)gk, (const uint16_t*)bk, cos, sin, text_len, eps, (int64_t)B * S * 2 * H, q_scale, k_sqmax};
    if (k_sqmax) {
        hipError_t me = hipMemsetAsync(k_sqmax, 0, sizeof(float) * (size_t)B * H, (hipStream_t)stream);
        if (me != hipSuccess) { tcx_set_error("tcx_qk_layernorm_rope: memset failed: %s", hipGetErrorString(me)); return (int)me; }
    }
    if (H == 48 && B <= 65535) {                         // the 5B model's head count: token-per-wave kernel
        // tokens per wave: the kernel holds 192 VGPRs -> 2 waves per SIMD = 8 resident waves (2 blocks) per CU.  With a fixed 8
        // tokens per wave the 2 x 17776 tokens of the product shape made 1112 blocks = 2.17 rounds of the 512 resident ones (a
        // third round 17 % full: 28 % of the kernel idle).  Size the waves so that ALL of them are resident at once (one round,
        // everybody finishes together); never fewer than 8 tokens (the per-wave parameter loads amortise over them).
        const uint32_t ncu = tcx_cu_count();                  // cached per device; 0 (query failed) -> the MI355X's 256
        const int64_t resident_waves = (int64_t)(ncu > 0 ? ncu : 256) * 8;
        int64_t tpw = ((int64_t)B * S + resident_waves - 1) / resident_waves;
        // per-batch rounding: blocks are per batch item (blockIdx.y), 4 waves each
        while (tpw < S && (int64_t)B * ((S + 4 * tpw - 1) / (4 * tpw)) * 4 > resident_waves) ++tpw;
        if (tpw < 8) tpw = 8;
        p.tokens_per_wave = (int32_t)tpw;
        const int per_block = 4 * (int)tpw;
        hipLaunchKernelGGL(qk_ln_rope_tok_kernel<6>, dim3((unsigned)((S + per_block - 1) / per_block), (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, p);
        TCX_LAUNCH_RET();
    }
    const int64_t nblk = (p.nvec + 31) / 32;
    TCX_CHECK(nblk < (1ll << 31), TCX_E_SHAPE, "tcx_qk_layernorm_rope: grid too large");
    hipLaunchKernelGGL(qk_ln_rope_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    TCX_LAUNCH_RET();
}
