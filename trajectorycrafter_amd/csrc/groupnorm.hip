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

__global__ __launch_bounds__(256) void gn_apply_kernel(const ApplyParams p) {
    const int cpr = p.C >> 3, cpg = p.C / p.G;
    const int64_t S = (int64_t)p.T * p.H * p.W;
    const int64_t nchunk = (int64_t)p.N * S * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        const int64_t row = i / cpr;             // (n, t, y, x)
        const int n = (int)(row / S);
        float v[8], g[8], b[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(p.x + i * 8), v);
        unpack8(*reinterpret_cast<const u32x4*>(p.gw + 8 * ch), g);
        unpack8(*reinterpret_cast<const u32x4*>(p.gb + 8 * ch), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int grp = (8 * ch + e) / cpg;
            const float mean = p.stats[2 * (n * p.G + grp)], rstd = p.stats[2 * (n * p.G + grp) + 1];
            o[e] = (v[e] - mean) * rstd * g[e] + b[e];
        }
        if (p.ytab) {
            const int64_t rs = row - (int64_t)n * S;
            const int t = (int)(rs / ((int64_t)p.H * p.W));
            const int rem = (int)(rs - (int64_t)t * p.H * p.W);
            const int yy = rem / p.W, xx = rem - yy * p.W;
            const int zt = p.ztm ? p.ztm[t] : t;
            const int zy = (int)(((int64_t)yy * p.Hz) / p.H), zx = (int)(((int64_t)xx * p.Wz) / p.W);
            const int64_t zi = ((((int64_t)n * p.Tz + zt) * p.Hz + zy) * p.Wz + zx) * p.C + 8 * ch;
            float ym[8], bm[8];
            unpack8(*reinterpret_cast<const u32x4*>(p.ytab + zi), ym);
            unpack8(*reinterpret_cast<const u32x4*>(p.btab + zi), bm);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(o[e], ym[e], bm[e]);
        }
        if (p.silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] / (1.0f + __expf(-o[e]));
        }
        *reinterpret_cast<u32x4*>(p.y + i * 8) = pack8(o);
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
    const int64_t nchunk = (int64_t)N * T * H * W * (C / 8);
    int64_t nb = (nchunk + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
    TCX_LAUNCH_RET();
}
