// K13: GroupNorm statistics and the fused GroupNorm * SpatialNorm-modulate + SiLU apply pass for the
// channels-last VAE activations (gfx950, HBM-bound).
//
// stats : two launches.  (1) each block streams a contiguous slab of rows with 16-byte loads and
//         accumulates per-CHANNEL shifted sums  sum(x - K_c), sum((x - K_c)^2)  (K_c = x[n,0,c], a
//         pivot that removes the cancellation of the one-pass variance) into `partial`;
//         (2) one thread per (n, group) combines slabs and channels in fp64 -> mean, rstd.
// apply : one 16-byte chunk (8 channels) per thread:  y = silu(GN(x) * Y[src] + Bt[src]) where
//         Y / Bt are the SpatialNorm 1x1x1 convs evaluated at the LOW resolution of zq (a nearest
//         resize commutes with a pointwise conv) and gathered with the nearest index, so the
//         resized zq and the two full-resolution conv outputs never exist in HBM.
#include "tcx_common.h"

namespace {

// partial layout: [N][nsplit][2][C] fp32
__global__ __launch_bounds__(256) void gn_partial_kernel(const uint16_t* x, float* partial, int64_t S, int32_t C, int32_t nsplit) {
    __shared__ float red[256][17];
    const int n = blockIdx.y, sp = blockIdx.x;
    const int cpr = C >> 3;                    // chunks per row (<= 256 enforced by the host)
    const int rows_per_it = 256 / cpr;
    const int tch = threadIdx.x % cpr, trow = threadIdx.x / cpr;
    const bool act = trow < rows_per_it;
    const int64_t rows_per_split = (S + nsplit - 1) / nsplit;
    const int64_t r0 = (int64_t)sp * rows_per_split;
    const int64_t r1 = r0 + rows_per_split < S ? r0 + rows_per_split : S;
    const uint16_t* xn = x + (int64_t)n * S * C;
    float piv[8], s1[8], s2[8];
    unpack8(*reinterpret_cast<const u32x4*>(xn + 8 * tch), piv);
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (act) {
        int64_t row = r0 + trow;
        // four independent 16-byte loads in flight per thread (the loop is latency bound otherwise)
        for (; row + 3 * (int64_t)rows_per_it < r1; row += 4 * (int64_t)rows_per_it) {
            u32x4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const u32x4*>(xn + (row + (int64_t)u * rows_per_it) * C + 8 * tch);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
                unpack8(raw[u], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[e] - piv[e];
                    s1[e] += d;
                    s2[e] = __builtin_fmaf(d, d, s2[e]);
                }
            }
        }
        for (; row < r1; row += rows_per_it) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(xn + row * C + 8 * tch), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[e] - piv[e];
                s1[e] += d;
                s2[e] = __builtin_fmaf(d, d, s2[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[threadIdx.x][e] = s1[e];
        red[threadIdx.x][8 + e] = s2[e];
    }
    __syncthreads();
    // threads 0 .. cpr-1 fold the rows_per_it partials of their chunk
    if (threadIdx.x < cpr) {
        float a[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0.f;
        for (int j = 0; j < rows_per_it; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] += red[j * cpr + threadIdx.x][e];
        float* out = partial + (((int64_t)n * nsplit + sp) * 2) * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            out[8 * threadIdx.x + e] = a[e];
            out[C + 8 * threadIdx.x + e] = a[8 + e];
        }
    }
}

// one 256-thread block per (n, group): fp64 combine of the slabs and of the group's channels
__global__ __launch_bounds__(256) void gn_finalize_kernel(const uint16_t* x, const float* partial, float* stats, int64_t S, int32_t C,
                                                          int32_t G, int32_t nsplit, float eps) {
    __shared__ double r1[256], r2[256];
    __shared__ double cs1[64], cs2[64];
    const int n = blockIdx.x / G, g = blockIdx.x - n * G;
    const int cpg = C / G;                      // <= 64 (host check)
    const int tc = threadIdx.x % cpg, ts = threadIdx.x / cpg, nts = 256 / cpg;
    double a1 = 0.0, a2 = 0.0;
    if (ts < nts) {
        const int c = g * cpg + tc;
        for (int sp = ts; sp < nsplit; sp += nts) {
            const float* pp = partial + (((int64_t)n * nsplit + sp) * 2) * C;
            a1 += pp[c];
            a2 += pp[C + c];
        }
    }
    r1[threadIdx.x] = a1;
    r2[threadIdx.x] = a2;
    __syncthreads();
    if (threadIdx.x < cpg) {
        double b1 = 0.0, b2 = 0.0;
        for (int j = 0; j < nts; ++j) {
            b1 += r1[j * cpg + threadIdx.x];
            b2 += r2[j * cpg + threadIdx.x];
        }
        cs1[threadIdx.x] = b1;
        cs2[threadIdx.x] = b2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double cnt = (double)S;
        double tot = 0.0;
        for (int j = 0; j < cpg; ++j) tot += cs1[j] + cnt * (double)bf16_bits_to_f32(x[(int64_t)n * S * C + g * cpg + j]);
        const double mean = tot / (cnt * cpg);
        double m2 = 0.0;           // sum (x - mean)^2 = sum (d - (mean - K))^2 with d = x - K
        for (int j = 0; j < cpg; ++j) {
            const double dm = mean - (double)bf16_bits_to_f32(x[(int64_t)n * S * C + g * cpg + j]);
            m2 += cs2[j] - 2.0 * dm * cs1[j] + cnt * dm * dm;
        }
        double var = m2 / (cnt * cpg);
        if (var < 0.0) var = 0.0;
        stats[2 * blockIdx.x] = (float)mean;
        stats[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

struct ApplyParams {
    const uint16_t* x;
    uint16_t* y;
    const float* stats;
    const uint16_t *gw, *gb, *ytab, *btab;
    const int32_t* ztm;
    int32_t N, T, H, W, C, G, Tz, Hz, Wz, silu;
};

// One workgroup per image row (n, t, y); thread -> (16-byte channel chunk ch = tid % cpr, pixel tid / cpr + k * ppp).
// Everything that does not depend on the pixel is hoisted: the chunk's 8 (mean, rstd * gamma, beta) triples live in registers,
// the zq frame / row of the SpatialNorm tables is workgroup-uniform, and the nearest zq column x * Wz / W advances by a
// Bresenham step (quotient + remainder) instead of a division per pixel.  Per 16-byte chunk: ~60 VALU + 16 transcendental
// against the ~400 lane-operations a CU can afford per 32 bytes of HBM traffic, so the kernel is HBM-bound (the previous form,
// one flat index per chunk decoded with six 64-bit divisions, per-element group lookups from global memory and an IEEE
// division per SiLU, ran at 1.1 TB/s).  UNROLL independent pixels per thread keep enough loads in flight.
template <bool ZQ, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const ApplyParams p) {
    constexpr int UNROLL = 4;
    const int cpr = p.C >> 3, cpg = p.C / p.G;
    const int ppp = 256 / cpr;                            // pixels per pass (cpr <= 256: host check)
    const int ch = threadIdx.x % cpr, px0 = threadIdx.x / cpr;
    if (px0 >= ppp) return;                               // cpr does not divide 256: the remainder threads idle
    const int rowid = blockIdx.x;                         // (n * T + t) * H + y
    const int y = rowid % p.H, nt = rowid / p.H;
    const int t = nt % p.T, n = nt / p.T;
    float mean[8], sc[8], sh[8];
    {
        float g[8], b[8];
        unpack8(*reinterpret_cast<const u32x4*>(p.gw + 8 * ch), g);
        unpack8(*reinterpret_cast<const u32x4*>(p.gb + 8 * ch), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int grp = (8 * ch + e) / cpg;
            mean[e] = p.stats[2 * (n * p.G + grp)];
            sc[e] = p.stats[2 * (n * p.G + grp) + 1] * g[e];
            sh[e] = b[e];
        }
    }
    const uint16_t* xr = p.x + (int64_t)rowid * p.W * p.C + 8 * ch;
    uint16_t* yr = p.y + (int64_t)rowid * p.W * p.C + 8 * ch;
    const uint16_t *yt = nullptr, *bt = nullptr;
    int zq = 0, zr = 0, dq = 0, dr = 0;                   // x * Wz = zq * W + zr for this thread's current pixel; step per pass
    if constexpr (ZQ) {
        const int zt = p.ztm ? p.ztm[t] : t;
        const int zy = (int)(((int64_t)y * p.Hz) / p.H);
        const int64_t zrow = (((int64_t)n * p.Tz + zt) * p.Hz + zy) * p.Wz;
        yt = p.ytab + zrow * p.C + 8 * ch;
        bt = p.btab + zrow * p.C + 8 * ch;
        zq = (px0 * p.Wz) / p.W; zr = px0 * p.Wz - zq * p.W;
        dq = (ppp * p.Wz) / p.W; dr = ppp * p.Wz - dq * p.W;
    }
    auto one = [&](const u32x4 raw, const u32x4 yraw, const u32x4 braw) __attribute__((always_inline)) -> u32x4 {
        float v[8], o[8];
        unpack8(raw, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(v[e] - mean[e], sc[e], sh[e]);
        if constexpr (ZQ) {
            float ym[8], bm[8];
            unpack8(yraw, ym);
            unpack8(braw, bm);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(o[e], ym[e], bm[e]);
        }
        if constexpr (SILU) {                             // x / (1 + 2^(-x log2 e)): one exp2 + one rcp (result rounded to bf16)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * o[e]));
        }
        return pack8(o);
    };
    int px = px0;
    for (; px + (UNROLL - 1) * ppp < p.W; px += UNROLL * ppp) {
        u32x4 raw[UNROLL], yraw[UNROLL], braw[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            raw[u] = *reinterpret_cast<const u32x4*>(xr + (int64_t)(px + u * ppp) * p.C);
            if constexpr (ZQ) {
                yraw[u] = *reinterpret_cast<const u32x4*>(yt + (int64_t)zq * p.C);
                braw[u] = *reinterpret_cast<const u32x4*>(bt + (int64_t)zq * p.C);
                zq += dq; zr += dr;
                if (zr >= p.W) { zr -= p.W; ++zq; }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) *reinterpret_cast<u32x4*>(yr + (int64_t)(px + u * ppp) * p.C) = one(raw[u], yraw[u], braw[u]);
    }
    for (; px < p.W; px += ppp) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(xr + (int64_t)px * p.C);
        u32x4 yraw = {0, 0, 0, 0}, braw = {0, 0, 0, 0};
        if constexpr (ZQ) {
            yraw = *reinterpret_cast<const u32x4*>(yt + (int64_t)zq * p.C);
            braw = *reinterpret_cast<const u32x4*>(bt + (int64_t)zq * p.C);
            zq += dq; zr += dr;
            if (zr >= p.W) { zr -= p.W; ++zq; }
        }
        *reinterpret_cast<u32x4*>(yr + (int64_t)px * p.C) = one(raw, yraw, braw);
    }
}

}  // namespace

extern "C" int tcx_groupnorm_stats(const void* x, float* stats, float* partial, int32_t N, int64_t spatial, int32_t C,
                                   int32_t G, float eps, int32_t nsplit, void* stream) {
    TCX_CHECK(x && stats && partial, TCX_E_NULL, "tcx_groupnorm_stats: null pointer");
    TCX_CHECK(N > 0 && N < 65536 && spatial > 0 && C > 0 && G > 0 && C % G == 0 && C % 8 == 0 && C <= 2048, TCX_E_SHAPE,
              "tcx_groupnorm_stats: need C %% G == 0, C %% 8 == 0, C <= 2048 (C=%d G=%d)", C, G);
    TCX_CHECK(nsplit > 0 && nsplit <= 65535, TCX_E_SHAPE, "tcx_groupnorm_stats: bad nsplit %d", nsplit);
    TCX_CHECK(C / G <= 64, TCX_E_SHAPE, "tcx_groupnorm_stats: at most 64 channels per group (C=%d G=%d)", C, G);
    TCX_CHECK(tcx_aligned16(x), TCX_E_ALIGN, "tcx_groupnorm_stats: x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nsplit, N), dim3(256), 0, st, (const uint16_t*)x, partial, spatial, C, nsplit);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(256), 0, st, (const uint16_t*)x, partial, stats, spatial, C, G, nsplit, eps);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_groupnorm_spatialnorm_silu(const void* x, void* y, const float* stats, const void* gn_w, const void* gn_b,
                                              const void* ytab, const void* btab, int32_t N, int32_t T, int32_t H, int32_t W,
                                              int32_t C, int32_t G, int32_t Tz, int32_t Hz, int32_t Wz, const int32_t* z_t_map,
                                              int32_t apply_silu, void* stream) {
    TCX_CHECK(x && y && stats && gn_w && gn_b, TCX_E_NULL, "tcx_groupnorm_spatialnorm_silu: null pointer");
    TCX_CHECK((ytab != nullptr) == (btab != nullptr), TCX_E_NULL, "tcx_groupnorm_spatialnorm_silu: ytab and btab go together");
    TCX_CHECK(N > 0 && T > 0 && H > 0 && W > 0 && C > 0 && G > 0 && C % G == 0 && C % 8 == 0, TCX_E_SHAPE, "tcx_groupnorm_spatialnorm_silu: bad shape");
    if (ytab) TCX_CHECK(Tz > 0 && Hz > 0 && Wz > 0 && Hz <= H && Wz <= W, TCX_E_SHAPE, "tcx_groupnorm_spatialnorm_silu: bad zq grid");
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(y) && tcx_aligned16(gn_w) && tcx_aligned16(gn_b) && tcx_aligned16(ytab) && tcx_aligned16(btab),
              TCX_E_ALIGN, "tcx_groupnorm_spatialnorm_silu: pointers must be 16-byte aligned");
    ApplyParams p{(const uint16_t*)x, (uint16_t*)y, stats, (const uint16_t*)gn_w, (const uint16_t*)gn_b, (const uint16_t*)ytab,
                  (const uint16_t*)btab, z_t_map, N, T, H, W, C, G, Tz, Hz, Wz, apply_silu};
    TCX_CHECK(C <= 2048 && (int64_t)N * T * H < (1ll << 31) && (int64_t)W * Wz < (1ll << 30), TCX_E_SHAPE,
              "tcx_groupnorm_spatialnorm_silu: C <= 2048 and N * T * H < 2^31 required (C=%d)", C);
    const dim3 grid((unsigned)((int64_t)N * T * H)), block(256);      // one workgroup per image row
    hipStream_t st = (hipStream_t)stream;
    if (ytab) {
        if (apply_silu) hipLaunchKernelGGL((gn_apply_kernel<true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gn_apply_kernel<true, false>), grid, block, 0, st, p);
    } else {
        if (apply_silu) hipLaunchKernelGGL((gn_apply_kernel<false, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gn_apply_kernel<false, false>), grid, block, 0, st, p);
    }
    TCX_LAUNCH_RET();
}
