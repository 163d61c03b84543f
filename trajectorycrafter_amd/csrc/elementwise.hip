// HBM-bound elementwise / gather kernels (gfx950): K5 gated residual, K6 bias+GELU(tanh) epilogue,
// K8 patchify / unpatchify, K10 CFG + DDIM step, K14 layout changes.  16-byte vector accesses,
// fp32 arithmetic, one rounding at the store; grids capped and grid-strided.
#include "tcx_common.h"

namespace {

constexpr int kMaxBlocks = 256 * 8;

inline unsigned grid_for(int64_t nvec, int threads = 256) {
    int64_t b = (nvec + threads - 1) / threads;
    if (b > kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// x[b, r, :] += gate[b, :] * y[b, r, :]
__global__ __launch_bounds__(256) void gated_residual_kernel(uint16_t* x, const uint16_t* y, int32_t rows, int32_t C,
                                                            int64_t xsb, int64_t ysb, const uint16_t* gate_v,
                                                            const uint16_t* gate_t, int64_t gsb, int32_t text_len) {
    const int b = blockIdx.y;
    const int cpr = C >> 3;                       // chunks per row
    const int64_t nchunk = (int64_t)rows * cpr;
    uint16_t* xb = x + (int64_t)b * xsb;
    const uint16_t* yb = y + (int64_t)b * ysb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / cpr), ch = (int)(i - (int64_t)row * cpr);
        float xv[8], yv[8], g[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(xb + i * 8), xv);
        unpack8(*reinterpret_cast<const u32x4*>(yb + i * 8), yv);
        const uint16_t* gp = row < text_len ? gate_t : gate_v;
        if (gp) {
            unpack8(*reinterpret_cast<const u32x4*>(gp + (int64_t)b * gsb + 8 * ch), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(g[e], yv[e], xv[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = xv[e] + yv[e];
        }
        *reinterpret_cast<u32x4*>(xb + i * 8) = pack8(o);
    }
}

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))),  tanh(u) = 1 - 2 / (1 + e^{2u})
    const float u = 0.7978845608028654f * __builtin_fmaf(0.044715f * x * x, x, x);
    const float e = __expf(2.0f * u);
    const float t = 1.0f - 2.0f / (1.0f + e);
    return 0.5f * x * (1.0f + t);
}

__global__ __launch_bounds__(256) void bias_gelu_kernel(const uint16_t* x, const uint16_t* bias, uint16_t* y,
                                                        int64_t nchunk, int32_t cpr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8], bb[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), v);
        if (bias) {
            unpack8(*reinterpret_cast<const u32x4*>(bias + 8 * (i % cpr)), bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bb[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gelu_tanh(v[e]);
        *reinterpret_cast<u32x4*>(y + i * 8) = pack8(o);
    }
}

template <int OP>  // 0: scale, 1: silu
__global__ __launch_bounds__(256) void unary_kernel(const uint16_t* x, uint16_t* y, int64_t n, float s) {
    const int64_t nchunk = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = OP == 0 ? v[e] * s : v[e] / (1.0f + __expf(-v[e]));
        *reinterpret_cast<u32x4*>(y + i * 8) = pack8(o);
    }
    // ragged tail (n % 8) handled by the first threads
    const int64_t tail0 = nchunk << 3;
    const int64_t t = tail0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float v = bf16_bits_to_f32(x[t]);
        const float o = OP == 0 ? v * s : v / (1.0f + __expf(-v));
        y[t] = (uint16_t)(pack_bf16(o, 0.f) & 0xffff);
    }
}

// y[b,s,h,:] = bf16(x[b,s,h,:] * scale) and sqmax[b,h] = max_s |y[b,s,h,:]|^2 (of the ROUNDED values): the k * scale of the
// cross-attention plus the Cauchy-Schwarz input of the bound-centred attention loop.  D / 8 lanes per head vector.
// A wave owns 64 / LPV heads of one batch item and walks kTokPerWave rows: the running max stays in registers and
// costs one atomic per head and wave (per-vector atomics on B*H addresses made the first version 7x slower than the copy).
constexpr int kTokPerWave = 32;
template <int LPV>
__global__ __launch_bounds__(256) void scale_sqmax_kernel(const uint16_t* x, uint16_t* y, int32_t S, int32_t H, int64_t xsb,
                                                          int64_t xss, float scale, float* sqmax) {
    constexpr int HPW = 64 / LPV;                       // heads per wave
    const int lane = threadIdx.x & 63, sub = lane % LPV, hl = lane / LPV;
    const int hgroups = (H + HPW - 1) / HPW;
    const int b = blockIdx.z, hg = blockIdx.y;
    const int h = hg * HPW + hl;
    const bool hok = h < H;
    const int s0 = ((int)blockIdx.x * 4 + (threadIdx.x >> 6)) * kTokPerWave;
    (void)hgroups;
    float best = 0.f;
    const int hc = hok ? h : H - 1;
    const uint16_t* xb = x + (int64_t)b * xsb + (int64_t)hc * (LPV * 8) + 8 * sub;
    for (int sbase = s0; sbase < S && sbase < s0 + kTokPerWave; sbase += 4) {      // four rows in flight per lane
        u32x4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const u32x4*>(xb + (int64_t)min(sbase + u, S - 1) * xss);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int srow = sbase + u;
            float v[8], o[8], r[8];
            unpack8(raw[u], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[e] * scale;
            const u32x4 packed = pack8(o);
            if (hok && srow < S) *reinterpret_cast<u32x4*>(y + (((int64_t)b * S + srow) * H + h) * (LPV * 8) + 8 * sub) = packed;
            unpack8(packed, r);
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(r[e], r[e], ss);
            ss = LPV == 8 ? group8_sum(ss) : group16_sum(ss);
            if (srow < S) best = fmaxf(best, ss);
        }
    }
    if (hok && sub == 0 && s0 < S) {
        unsigned* dst = reinterpret_cast<unsigned*>(sqmax + (int64_t)b * H + h);
        const unsigned bits = __float_as_uint(best);          // non-negative floats order like their bit patterns
        if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
}

// out[(b,f,gy,gx), (c,ky,kx)] = in[b,f,c, gy*p+ky, gx*p+kx], c over the concat of a (Ca ch) and b (Cb ch); rows are
// ks >= K elements long, zero-filled past K (the GEMM wants K % 128 == 0)
__global__ __launch_bounds__(256) void patchify_kernel(const uint16_t* a, const uint16_t* bsrc, uint16_t* out,
                                                       int32_t BF, int32_t Ca, int32_t Cb, int32_t H, int32_t W, int32_t p, int32_t ks) {
    const int gh = H / p, gw = W / p, Ct = Ca + Cb, K = Ct * p * p;
    const int64_t n = (int64_t)BF * gh * gw * ks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i % ks);
        const int64_t m = i / ks;
        if (kk >= K) { out[i] = 0; continue; }
        const int gx = (int)(m % gw), gy = (int)((m / gw) % gh);
        const int64_t bf = m / ((int64_t)gw * gh);
        const int kx = kk % p, ky = (kk / p) % p, c = kk / (p * p);
        const int yy = gy * p + ky, xx = gx * p + kx;
        uint16_t v;
        if (c < Ca) v = a[((bf * Ca + c) * H + yy) * W + xx];
        else v = bsrc[((bf * Cb + (c - Ca)) * H + yy) * W + xx];
        out[i] = v;
    }
}

// out[b,f,c, gy*p+py, gx*p+px] = x[b, (f,gy,gx), (c,py,px)]
template <bool F32>
__global__ __launch_bounds__(256) void unpatchify_kernel(const uint16_t* x, void* out, int32_t BF, int32_t C, int32_t H,
                                                         int32_t W, int32_t p) {
    const int gh = H / p, gw = W / p;
    const int64_t n = (int64_t)BF * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t bf = i / ((int64_t)W * H * C);
        const int gy = yy / p, py = yy % p, gx = xx / p, px = xx % p;
        const uint16_t v = x[((bf * gh + gy) * gw + gx) * ((int64_t)C * p * p) + (c * p + py) * p + px];
        if constexpr (F32) reinterpret_cast<float*>(out)[i] = bf16_bits_to_f32(v);
        else reinterpret_cast<uint16_t*>(out)[i] = v;
    }
}

template <bool PRED_F32>
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const void* u, const void* c, const uint16_t* x, uint16_t* out,
                                                       int64_t n, float g, float sa, float sb, float sap, float sbp,
                                                       const float* vnoise = nullptr, float std_dev = 0.f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float uu, cc = 0.f;
        if constexpr (PRED_F32) {
            uu = reinterpret_cast<const float*>(u)[i];
            if (c) cc = reinterpret_cast<const float*>(c)[i];
        } else {
            uu = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(u)[i]);
            if (c) cc = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(c)[i]);
        }
        const float noise = c ? uu + g * (cc - uu) : uu;
        const float xs = bf16_bits_to_f32(x[i]);
        // 0-dim fp32 scalar * bf16 tensor stays bf16 in the reference (SURVEY 8c promotion quirk)
        const float x0 = round_bf16(sa * xs) - sb * noise;
        const float eps = sa * noise + round_bf16(sb * xs);
        float prev = sap * x0 + sbp * eps;
        if (vnoise) prev = prev + std_dev * vnoise[i];          // eta > 0: sbp is sqrt(1 - a_prev - std^2), + std * N(0, 1)
        out[i] = (uint16_t)(pack_bf16(prev, 0.f) & 0xffff);
    }
}

// CogVideoXDDIMScheduler.step (sampler "DDIM_Cog"): x0 = bf16r(sa x) - sb noise;  prev = bf16r(ca x) + cb x0
template <bool PRED_F32>
__global__ __launch_bounds__(256) void cfg_ddim_cog_kernel(const void* u, const void* c, const uint16_t* x, uint16_t* out,
                                                           int64_t n, float g, float sa, float sb, float ca, float cb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float uu, cc = 0.f;
        if constexpr (PRED_F32) {
            uu = reinterpret_cast<const float*>(u)[i];
            if (c) cc = reinterpret_cast<const float*>(c)[i];
        } else {
            uu = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(u)[i]);
            if (c) cc = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(c)[i]);
        }
        const float noise = c ? uu + g * (cc - uu) : uu;
        const float xs = bf16_bits_to_f32(x[i]);
        const float x0 = round_bf16(sa * xs) - sb * noise;      // 0-dim fp64 scalar * bf16 tensor stays bf16 (same promotion quirk)
        const float prev = round_bf16(ca * xs) + cb * x0;
        out[i] = (uint16_t)(pack_bf16(prev, 0.f) & 0xffff);
    }
}

// Sigma-parametrised samplers (diffusers EulerDiscrete / EulerAncestralDiscrete / DPMSolverMultistep++ 2M), v-prediction.
// The coefficients are the library's own 0-dim fp32 scalars, computed on the host; every per-element operation below is the
// fp32 operation the library's tensor expression performs, in its order (no contraction: -ffp-contract=off).
//   KIND 0 (Euler):  x0 = v a + x / s2p1;  d = (x - x0) / sigma;  prev = x + d dt  [+ noise sigma_up]
//   KIND 1 (DPM++):  x0 = bf16r(alpha x) - sig v (kept in `hist_out`);  prev = A x - B x0  [- (B/2) (inv_r0 (x0 - x0_prev))]
template <bool PRED_F32, int KIND>
__global__ __launch_bounds__(256) void cfg_sigma_step_kernel(const void* u, const void* c, const uint16_t* x, uint16_t* out,
                                                             int64_t n, float g, float k0, float k1, float k2, float k3, float k4,
                                                             const float* hist_in, float* hist_out, const float* noise) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float uu, cc = 0.f;
        if constexpr (PRED_F32) {
            uu = reinterpret_cast<const float*>(u)[i];
            if (c) cc = reinterpret_cast<const float*>(c)[i];
        } else {
            uu = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(u)[i]);
            if (c) cc = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(c)[i]);
        }
        const float v = c ? uu + g * (cc - uu) : uu;
        const float xs = bf16_bits_to_f32(x[i]);
        float prev;
        if constexpr (KIND == 0) {
            const float x0 = v * k0 + xs / k1;
            const float d = (xs - x0) / k2;
            prev = xs + d * k3;
            if (noise) prev = prev + noise[i] * k4;
        } else {
            const float x0 = round_bf16(k0 * xs) - k1 * v;
            hist_out[i] = x0;
            prev = k2 * xs - k3 * x0;
            if (hist_in) prev = prev - (0.5f * k3) * (k4 * (x0 - hist_in[i]));
        }
        out[i] = (uint16_t)(pack_bf16(prev, 0.f) & 0xffff);
    }
}

// PNDM (diffusers PNDMScheduler, v-prediction, Runge-Kutta warm-up + 4th-order linear multistep).  mo = guided model output (fp32).
//   MODE 0 / 1 (Runge-Kutta evaluation 1 / 2-3 of a group): cur_out = cur_in + w mo (w = 1/6, 1/3; cur_in null = the integer 0 the
//            library starts from: 0 + w mo);  eff = mo;  MODE 0 also stores mo to `mo_out` (it joins the multistep history)
//   MODE 2 (evaluation 4): eff = cur_in + w mo (w = 1/6)
//   MODE 3 (multistep): eff = (1/24) (((55 mo - 59 e1) + 37 e2) - 9 e3);  mo stored to `mo_out`
//   then  eps = sa eff + bf16r(sb x);  prev = bf16r(sc x) - (diff eps) / denom   (x = the bf16 sample `_get_prev_sample` is given)
template <bool PRED_F32>
__global__ __launch_bounds__(256) void cfg_pndm_step_kernel(const void* u, const void* c, const uint16_t* x, uint16_t* out, int64_t n,
                                                            float g, int mode, float w, float sa, float sb, float sc, float diff,
                                                            float denom, const float* e1, const float* e2, const float* e3,
                                                            const float* cur_in, float* cur_out, float* mo_out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float uu, cc = 0.f;
        if constexpr (PRED_F32) {
            uu = reinterpret_cast<const float*>(u)[i];
            if (c) cc = reinterpret_cast<const float*>(c)[i];
        } else {
            uu = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(u)[i]);
            if (c) cc = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(c)[i]);
        }
        const float mo = c ? uu + g * (cc - uu) : uu;
        float eff = mo;
        if (mode <= 1) {
            const float t = w * mo;
            cur_out[i] = cur_in ? cur_in[i] + t : 0.f + t;
            if (mode == 0) mo_out[i] = mo;
        } else if (mode == 2) {
            eff = cur_in[i] + w * mo;
        } else {
            mo_out[i] = mo;
            eff = (1.0f / 24.0f) * (((55.0f * mo - 59.0f * e1[i]) + 37.0f * e2[i]) - 9.0f * e3[i]);
        }
        const float xs = bf16_bits_to_f32(x[i]);
        const float eps = sa * eff + round_bf16(sb * xs);
        const float prev = round_bf16(sc * xs) - (diff * eps) / denom;
        out[i] = (uint16_t)(pack_bf16(prev, 0.f) & 0xffff);
    }
}

// y = bf16(x / d): `scale_model_input` of the Euler samplers (a bf16 tensor divided by a 0-dim fp32 tensor stays bf16)
__global__ __launch_bounds__(256) void div_kernel(const uint16_t* x, uint16_t* y, int64_t n, float d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (uint16_t)(pack_bf16(bf16_bits_to_f32(x[i]) / d, 0.f) & 0xffff);
}

// [N, C, S] (S = T*H*W) bf16 -> channels-last [N, S, C] bf16, scaled by `mul`; LDS-tiled transpose
__global__ __launch_bounds__(256) void ncthw_to_cl_kernel(const uint16_t* x, uint16_t* y, int32_t C, int64_t S, float mul) {
    __shared__ uint16_t tile[32][33];
    const int n = blockIdx.z;
    const int64_t s0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const uint16_t* xn = x + (int64_t)n * C * S;
    uint16_t* yn = y + (int64_t)n * C * S;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const int64_t s = s0 + tx;
        tile[j][tx] = (c < C && s < S) ? xn[(int64_t)c * S + s] : (uint16_t)0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t s = s0 + j;
        const int c = c0 + tx;
        if (c < C && s < S) {
            const float v = bf16_bits_to_f32(tile[tx][j]) * mul;
            yn[s * C + c] = (uint16_t)(pack_bf16(v, 0.f) & 0xffff);
        }
    }
}

// channels-last [N, S, C] bf16 -> frames fp32: y[n, c, out_offset + s] = clamp(x/2 + .5, 0, 1)
__global__ __launch_bounds__(256) void cl_to_frames_kernel(const uint16_t* x, float* y, int32_t C, int64_t S,
                                                           int64_t out_sstride, int64_t out_off) {
    const int n = blockIdx.y;
    const int64_t total = S * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / C;
        const int c = (int)(i - s * C);
        // bf16 rounding of (x/2 + 0.5) first: the reference computes it in the VAE dtype (bf16)
        float v = round_bf16(bf16_bits_to_f32(x[(int64_t)n * total + i]) * 0.5f + 0.5f);
        v = fminf(fmaxf(v, 0.f), 1.f);
        y[((int64_t)n * C + c) * out_sstride + out_off + s] = v;
    }
}

// temporal average pool (pairs; odd T keeps frame 0): x [N,T,S,C] -> y [N,T',S,C]
__global__ __launch_bounds__(256) void avgpool_t_kernel(const uint16_t* x, uint16_t* y, int32_t T, int32_t To, int64_t SC8) {
    const int n = blockIdx.y;
    const int odd = T & 1;
    const int64_t total = (int64_t)To * SC8;
    const uint16_t* xn = x + (int64_t)n * T * SC8 * 8;
    uint16_t* yn = y + (int64_t)n * To * SC8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int to = (int)(i / SC8);
        const int64_t r = i - (int64_t)to * SC8;
        float a[8], b[8], o[8];
        if (odd && to == 0) {
            *reinterpret_cast<u32x4*>(yn + i * 8) = *reinterpret_cast<const u32x4*>(xn + r * 8);
            continue;
        }
        const int t0 = odd ? 1 + 2 * (to - 1) : 2 * to;
        unpack8(*reinterpret_cast<const u32x4*>(xn + ((int64_t)t0 * SC8 + r) * 8), a);
        unpack8(*reinterpret_cast<const u32x4*>(xn + ((int64_t)(t0 + 1) * SC8 + r) * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (a[e] + b[e]) * 0.5f;
        *reinterpret_cast<u32x4*>(yn + i * 8) = pack8(o);
    }
}

// tile-seam blend of the tiled VAE decode: b[o, y, i] = a[o, y, i] * (1 - y/ext) + b[o, y, i] * (y/ext), y < ext, with the
// reference's eager bf16 roundings (each product, then the sum; the python-float weights enter at fp32)
__global__ __launch_bounds__(256) void blend_ramp_kernel(const uint16_t* a, uint16_t* b, int64_t outer, int32_t ext, int64_t inner,
                                                         int64_t a_so, int64_t a_se, int64_t b_so, int64_t b_se) {
    const int64_t per = (int64_t)ext * inner, total = outer * per;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = idx / per, r = idx - o * per;
        const int y = (int)(r / inner);
        const int64_t i = r - (int64_t)y * inner;
        const float wb = (float)((double)y / (double)ext), wa = (float)(1.0 - (double)y / (double)ext);
        uint16_t* bp = b + o * b_so + (int64_t)y * b_se + i;
        const float av = round_bf16(bf16_bits_to_f32(a[o * a_so + (int64_t)y * a_se + i]) * wa);
        const float bv = round_bf16(bf16_bits_to_f32(*bp) * wb);
        *bp = (uint16_t)(__float_as_uint(round_bf16(av + bv)) >> 16);
    }
}

}  // namespace

extern "C" int tcx_blend_ramp_bf16(const void* a, void* b, int64_t outer, int32_t ext, int64_t inner, int64_t a_outer_stride,
                                   int64_t a_ext_stride, int64_t b_outer_stride, int64_t b_ext_stride, void* stream) {
    TCX_CHECK(a && b, TCX_E_NULL, "tcx_blend_ramp_bf16: null pointer");
    TCX_CHECK(outer > 0 && ext > 0 && inner > 0, TCX_E_SHAPE, "tcx_blend_ramp_bf16: outer, ext and inner must be positive");
    TCX_CHECK(a_ext_stride >= inner && b_ext_stride >= inner && a_outer_stride >= (int64_t)ext * a_ext_stride &&
              b_outer_stride >= (int64_t)ext * b_ext_stride, TCX_E_SHAPE, "tcx_blend_ramp_bf16: strides shorter than the blended strip");
    hipLaunchKernelGGL(blend_ramp_kernel, dim3(grid_for(outer * ext * inner)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
                       (uint16_t*)b, outer, ext, inner, a_outer_stride, a_ext_stride, b_outer_stride, b_ext_stride);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_avgpool_t(const void* x, void* y, int32_t N, int32_t T, int64_t S, int32_t C, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_avgpool_t: null pointer");
    TCX_CHECK(N > 0 && N < 65536 && T > 0 && S > 0 && C > 0 && C % 8 == 0, TCX_E_SHAPE, "tcx_avgpool_t: bad shape (C %% 8 == 0 required)");
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(y), TCX_E_ALIGN, "tcx_avgpool_t: pointers must be 16-byte aligned");
    const int To = (T & 1) ? 1 + (T - 1) / 2 : T / 2;
    TCX_CHECK(To > 0, TCX_E_SHAPE, "tcx_avgpool_t: T must be >= 2 or odd");
    const int64_t SC8 = S * (C / 8);
    hipLaunchKernelGGL(avgpool_t_kernel, dim3(grid_for((int64_t)To * SC8), N), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (uint16_t*)y, T, To, SC8);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_gated_residual(void* x, const void* y, int32_t B, int32_t rows, int32_t C, int64_t xsb, int64_t ysb,
                                  const void* gate_v, const void* gate_t, int64_t gsb, int32_t text_len, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_gated_residual: null x/y");
    TCX_CHECK(B > 0 && B < 65536 && rows > 0 && C > 0 && C % 8 == 0, TCX_E_SHAPE, "tcx_gated_residual: bad shape B=%d rows=%d C=%d", B, rows, C);
    TCX_CHECK(xsb % 8 == 0 && ysb % 8 == 0 && gsb % 8 == 0, TCX_E_ALIGN, "tcx_gated_residual: strides must be multiples of 8");
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(y) && tcx_aligned16(gate_v) && tcx_aligned16(gate_t), TCX_E_ALIGN,
              "tcx_gated_residual: pointers must be 16-byte aligned");
    if (text_len > 0) TCX_CHECK((gate_t != nullptr) == (gate_v != nullptr), TCX_E_NULL, "tcx_gated_residual: gate_t must be given iff gate_v is");
    const int64_t nchunk = (int64_t)rows * (C / 8);
    dim3 grid(grid_for(nchunk), B);
    hipLaunchKernelGGL(gated_residual_kernel, grid, dim3(256), 0, (hipStream_t)stream, (uint16_t*)x, (const uint16_t*)y, rows, C,
                       xsb, ysb, (const uint16_t*)gate_v, (const uint16_t*)gate_t, gsb, text_len);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_bias_gelu_tanh(const void* x, const void* bias, void* y, int64_t rows, int32_t C, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_bias_gelu_tanh: null x/y");
    TCX_CHECK(rows > 0 && C > 0 && C % 8 == 0, TCX_E_SHAPE, "tcx_bias_gelu_tanh: bad shape rows=%lld C=%d", (long long)rows, C);
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(y) && tcx_aligned16(bias), TCX_E_ALIGN, "tcx_bias_gelu_tanh: pointers must be 16-byte aligned");
    const int64_t nchunk = rows * (C / 8);
    hipLaunchKernelGGL(bias_gelu_kernel, dim3(grid_for(nchunk)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (const uint16_t*)bias, (uint16_t*)y, nchunk, C / 8);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_scale_bf16(const void* x, void* y, int64_t n, float s, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_scale_bf16: null pointer");
    TCX_CHECK(n > 0, TCX_E_SHAPE, "tcx_scale_bf16: n must be positive");
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(y), TCX_E_ALIGN, "tcx_scale_bf16: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(unary_kernel<0>, dim3(grid_for((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n, s);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_scale_sqmax_bf16(const void* x, void* y, int32_t B, int32_t S, int32_t H, int32_t D, int64_t x_stride_b,
                                    int64_t x_stride_s, float scale, float* sqmax, void* stream) {
    TCX_CHECK(x && y && sqmax, TCX_E_NULL, "tcx_scale_sqmax_bf16: null pointer");
    TCX_CHECK(B > 0 && S > 0 && H > 0 && (D == 64 || D == 128), TCX_E_SHAPE, "tcx_scale_sqmax_bf16: bad shape B=%d S=%d H=%d D=%d", B, S, H, D);
    TCX_CHECK(x_stride_s >= (int64_t)H * D && x_stride_b % 8 == 0 && x_stride_s % 8 == 0, TCX_E_ALIGN, "tcx_scale_sqmax_bf16: bad strides");
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(y), TCX_E_ALIGN, "tcx_scale_sqmax_bf16: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipError_t me = hipMemsetAsync(sqmax, 0, sizeof(float) * (size_t)B * H, st);
    if (me != hipSuccess) { tcx_set_error("tcx_scale_sqmax_bf16: memset failed: %s", hipGetErrorString(me)); return (int)me; }
    const int lpv = D / 8, hpw = 64 / lpv;
    TCX_CHECK(B <= 65535 && (H + hpw - 1) / hpw <= 65535, TCX_E_SHAPE, "tcx_scale_sqmax_bf16: B / H exceed the grid limits");
    const dim3 grid((unsigned)((S + 4 * kTokPerWave - 1) / (4 * kTokPerWave)), (unsigned)((H + hpw - 1) / hpw), (unsigned)B);
    if (D == 64) hipLaunchKernelGGL(scale_sqmax_kernel<8>, grid, dim3(256), 0, st, (const uint16_t*)x, (uint16_t*)y, S, H, x_stride_b, x_stride_s, scale, sqmax);
    else hipLaunchKernelGGL(scale_sqmax_kernel<16>, grid, dim3(256), 0, st, (const uint16_t*)x, (uint16_t*)y, S, H, x_stride_b, x_stride_s, scale, sqmax);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_silu_bf16(const void* x, void* y, int64_t n, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_silu_bf16: null pointer");
    TCX_CHECK(n > 0, TCX_E_SHAPE, "tcx_silu_bf16: n must be positive");
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(y), TCX_E_ALIGN, "tcx_silu_bf16: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(unary_kernel<1>, dim3(grid_for((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n, 0.f);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_patchify(const void* a, const void* b, void* out, int32_t B, int32_t F, int32_t Ca, int32_t Cb,
                            int32_t H, int32_t W, int32_t p, int32_t k_stride, void* stream) {
    TCX_CHECK(a && out, TCX_E_NULL, "tcx_patchify: null pointer");
    TCX_CHECK((b != nullptr) == (Cb > 0), TCX_E_NULL, "tcx_patchify: b must be given iff Cb > 0");
    TCX_CHECK(B > 0 && F > 0 && Ca > 0 && Cb >= 0 && p > 0 && H % p == 0 && W % p == 0, TCX_E_SHAPE,
              "tcx_patchify: H=%d, W=%d must be divisible by patch %d", H, W, p);
    if (k_stride == 0) k_stride = (Ca + Cb) * p * p;
    TCX_CHECK(k_stride >= (Ca + Cb) * p * p, TCX_E_SHAPE, "tcx_patchify: k_stride=%d is shorter than a row (%d)", k_stride, (Ca + Cb) * p * p);
    const int64_t n = (int64_t)B * F * (H / p) * (W / p) * k_stride;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
                       (const uint16_t*)b, (uint16_t*)out, B * F, Ca, Cb, H, W, p, k_stride);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_unpatchify(const void* x, void* out, int32_t B, int32_t F, int32_t C, int32_t H, int32_t W, int32_t p,
                              int32_t out_dtype, void* stream) {
    TCX_CHECK(x && out, TCX_E_NULL, "tcx_unpatchify: null pointer");
    TCX_CHECK(B > 0 && F > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, TCX_E_SHAPE, "tcx_unpatchify: bad shape");
    TCX_CHECK(out_dtype == TCX_BF16 || out_dtype == TCX_F32, TCX_E_DTYPE, "tcx_unpatchify: bad out_dtype %d", out_dtype);
    const int64_t n = (int64_t)B * F * C * H * W;
    if (out_dtype == TCX_F32)
        hipLaunchKernelGGL(unpatchify_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, out, B * F, C, H, W, p);
    else
        hipLaunchKernelGGL(unpatchify_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, out, B * F, C, H, W, p);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_cfg_ddim_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance,
                                 float alpha_t, float alpha_prev, int32_t pred_dtype, void* stream) {
    TCX_CHECK(u && x && out, TCX_E_NULL, "tcx_cfg_ddim_step: null pointer");
    TCX_CHECK(n > 0, TCX_E_SHAPE, "tcx_cfg_ddim_step: n must be positive");
    TCX_CHECK(pred_dtype == TCX_BF16 || pred_dtype == TCX_F32, TCX_E_DTYPE, "tcx_cfg_ddim_step: bad pred_dtype %d", pred_dtype);
    TCX_CHECK(alpha_t >= 0.f && alpha_t <= 1.f && alpha_prev >= 0.f && alpha_prev <= 1.f, TCX_E_SHAPE, "tcx_cfg_ddim_step: alphas must be in [0,1]");
    const float sa = sqrtf(alpha_t), sb = sqrtf(1.0f - alpha_t), sap = sqrtf(alpha_prev), sbp = sqrtf(1.0f - alpha_prev);
    if (pred_dtype == TCX_F32)
        hipLaunchKernelGGL(cfg_ddim_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, sa, sb, sap, sbp);
    else
        hipLaunchKernelGGL(cfg_ddim_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, sa, sb, sap, sbp);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_cfg_ddim_eta_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance,
                                     float sqrt_alpha_t, float sqrt_beta_t, float sqrt_alpha_prev, float dir_coef, float std_dev,
                                     const float* variance_noise, int32_t pred_dtype, void* stream) {
    TCX_CHECK(u && x && out && variance_noise, TCX_E_NULL, "tcx_cfg_ddim_eta_step: null pointer");
    TCX_CHECK(n > 0, TCX_E_SHAPE, "tcx_cfg_ddim_eta_step: n must be positive");
    TCX_CHECK(pred_dtype == TCX_BF16 || pred_dtype == TCX_F32, TCX_E_DTYPE, "tcx_cfg_ddim_eta_step: bad pred_dtype %d", pred_dtype);
    TCX_CHECK(sqrt_alpha_t >= 0.f && sqrt_alpha_t <= 1.f && sqrt_beta_t >= 0.f && sqrt_beta_t <= 1.f && sqrt_alpha_prev >= 0.f &&
                  sqrt_alpha_prev <= 1.f && dir_coef >= 0.f && dir_coef <= 1.f && std_dev >= 0.f && std_dev <= 1.f, TCX_E_SHAPE,
              "tcx_cfg_ddim_eta_step: coefficients must be in [0, 1] (a NaN direction coefficient means eta too large for this step)");
    if (pred_dtype == TCX_F32)
        hipLaunchKernelGGL(cfg_ddim_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, c, (const uint16_t*)x, (uint16_t*)out, n,
                           guidance, sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, dir_coef, variance_noise, std_dev);
    else
        hipLaunchKernelGGL(cfg_ddim_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, c, (const uint16_t*)x, (uint16_t*)out, n,
                           guidance, sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, dir_coef, variance_noise, std_dev);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_cfg_ddim_cog_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance,
                                     float sqrt_alpha_t, float sqrt_beta_t, float coef_sample, float coef_x0, int32_t pred_dtype,
                                     void* stream) {
    TCX_CHECK(u && x && out, TCX_E_NULL, "tcx_cfg_ddim_cog_step: null pointer");
    TCX_CHECK(n > 0, TCX_E_SHAPE, "tcx_cfg_ddim_cog_step: n must be positive");
    TCX_CHECK(pred_dtype == TCX_BF16 || pred_dtype == TCX_F32, TCX_E_DTYPE, "tcx_cfg_ddim_cog_step: bad pred_dtype %d", pred_dtype);
    TCX_CHECK(sqrt_alpha_t >= 0.f && sqrt_alpha_t <= 1.f && sqrt_beta_t >= 0.f && sqrt_beta_t <= 1.f, TCX_E_SHAPE,
              "tcx_cfg_ddim_cog_step: sqrt(alpha_t), sqrt(1 - alpha_t) must be in [0,1]");
    if (pred_dtype == TCX_F32)
        hipLaunchKernelGGL(cfg_ddim_cog_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, sqrt_alpha_t, sqrt_beta_t, coef_sample, coef_x0);
    else
        hipLaunchKernelGGL(cfg_ddim_cog_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, sqrt_alpha_t, sqrt_beta_t, coef_sample, coef_x0);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_cfg_sigma_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance, int32_t kind,
                                  const float* coef, const float* hist_in, float* hist_out, const float* noise, int32_t pred_dtype,
                                  void* stream) {
    TCX_CHECK(u && x && out && coef, TCX_E_NULL, "tcx_cfg_sigma_step: null pointer");
    TCX_CHECK(n > 0, TCX_E_SHAPE, "tcx_cfg_sigma_step: n must be positive");
    TCX_CHECK(pred_dtype == TCX_BF16 || pred_dtype == TCX_F32, TCX_E_DTYPE, "tcx_cfg_sigma_step: bad pred_dtype %d", pred_dtype);
    TCX_CHECK(kind == TCX_STEP_EULER || kind == TCX_STEP_DPMPP_2M, TCX_E_SHAPE, "tcx_cfg_sigma_step: unknown kind %d", kind);
    const float k0 = coef[0], k1 = coef[1], k2 = coef[2], k3 = coef[3], k4 = coef[4];
    for (int j = 0; j < 5; ++j) TCX_CHECK(coef[j] == coef[j], TCX_E_SHAPE, "tcx_cfg_sigma_step: coef[%d] is NaN", j);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(n));
    if (kind == TCX_STEP_EULER) {
        TCX_CHECK(k1 > 0.f && k2 > 0.f, TCX_E_SHAPE, "tcx_cfg_sigma_step: Euler needs sigma^2 + 1 > 0 and sigma > 0 (coef[1], coef[2])");
        TCX_CHECK(!hist_in && !hist_out, TCX_E_SHAPE, "tcx_cfg_sigma_step: the Euler step keeps no history");
        if (pred_dtype == TCX_F32)
            hipLaunchKernelGGL((cfg_sigma_step_kernel<true, 0>), grid, dim3(256), 0, st, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, k0, k1, k2, k3, k4, nullptr, nullptr, noise);
        else
            hipLaunchKernelGGL((cfg_sigma_step_kernel<false, 0>), grid, dim3(256), 0, st, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, k0, k1, k2, k3, k4, nullptr, nullptr, noise);
    } else {
        TCX_CHECK(hist_out != nullptr, TCX_E_NULL, "tcx_cfg_sigma_step: DPM++ needs hist_out (the x0 prediction kept for the next step)");
        TCX_CHECK(!noise, TCX_E_SHAPE, "tcx_cfg_sigma_step: DPM-Solver++ (deterministic) takes no noise");
        TCX_CHECK(hist_in != hist_out, TCX_E_SHAPE, "tcx_cfg_sigma_step: hist_in and hist_out must be different buffers");
        if (pred_dtype == TCX_F32)
            hipLaunchKernelGGL((cfg_sigma_step_kernel<true, 1>), grid, dim3(256), 0, st, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, k0, k1, k2, k3, k4, hist_in, hist_out, nullptr);
        else
            hipLaunchKernelGGL((cfg_sigma_step_kernel<false, 1>), grid, dim3(256), 0, st, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, k0, k1, k2, k3, k4, hist_in, hist_out, nullptr);
    }
    TCX_LAUNCH_RET();
}

extern "C" int tcx_cfg_pndm_step(const void* u, const void* c, const void* x, void* out, int64_t n, float guidance, int32_t mode,
                                 const float* coef, const float* e1, const float* e2, const float* e3, const float* cur_in,
                                 float* cur_out, float* mo_out, int32_t pred_dtype, void* stream) {
    TCX_CHECK(u && x && out && coef, TCX_E_NULL, "tcx_cfg_pndm_step: null pointer");
    TCX_CHECK(n > 0, TCX_E_SHAPE, "tcx_cfg_pndm_step: n must be positive");
    TCX_CHECK(pred_dtype == TCX_BF16 || pred_dtype == TCX_F32, TCX_E_DTYPE, "tcx_cfg_pndm_step: bad pred_dtype %d", pred_dtype);
    TCX_CHECK(mode >= TCX_PNDM_PRK_FIRST && mode <= TCX_PNDM_PLMS4, TCX_E_SHAPE, "tcx_cfg_pndm_step: unknown mode %d", mode);
    for (int j = 0; j < 6; ++j) TCX_CHECK(coef[j] == coef[j], TCX_E_SHAPE, "tcx_cfg_pndm_step: coef[%d] is NaN", j);
    TCX_CHECK(coef[5] != 0.f, TCX_E_SHAPE, "tcx_cfg_pndm_step: the denominator (coef[5]) is zero");
    if (mode == TCX_PNDM_PRK_FIRST) TCX_CHECK(cur_out && mo_out, TCX_E_NULL, "tcx_cfg_pndm_step: first Runge-Kutta evaluation needs cur_out and mo_out");
    if (mode == TCX_PNDM_PRK_MID) TCX_CHECK(cur_in && cur_out, TCX_E_NULL, "tcx_cfg_pndm_step: middle Runge-Kutta evaluations need cur_in and cur_out");
    if (mode == TCX_PNDM_PRK_LAST) TCX_CHECK(cur_in != nullptr, TCX_E_NULL, "tcx_cfg_pndm_step: last Runge-Kutta evaluation needs cur_in");
    if (mode == TCX_PNDM_PLMS4) TCX_CHECK(e1 && e2 && e3 && mo_out, TCX_E_NULL, "tcx_cfg_pndm_step: the multistep update needs e1..e3 and mo_out");
    TCX_CHECK(mo_out == nullptr || (mo_out != e1 && mo_out != e2 && mo_out != e3), TCX_E_SHAPE, "tcx_cfg_pndm_step: mo_out must not alias the history");
    hipStream_t st = (hipStream_t)stream;
    if (pred_dtype == TCX_F32)
        hipLaunchKernelGGL(cfg_pndm_step_kernel<true>, dim3(grid_for(n)), dim3(256), 0, st, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, mode,
                           coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], e1, e2, e3, cur_in, cur_out, mo_out);
    else
        hipLaunchKernelGGL(cfg_pndm_step_kernel<false>, dim3(grid_for(n)), dim3(256), 0, st, u, c, (const uint16_t*)x, (uint16_t*)out, n, guidance, mode,
                           coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], e1, e2, e3, cur_in, cur_out, mo_out);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_div_bf16(const void* x, void* y, int64_t n, float d, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_div_bf16: null pointer");
    TCX_CHECK(n > 0 && d != 0.f && d == d, TCX_E_SHAPE, "tcx_div_bf16: n must be positive and the divisor a non-zero number");
    hipLaunchKernelGGL(div_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n, d);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_ncthw_to_cl(const void* x, void* y, int32_t N, int32_t C, int64_t spatial, float mul, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_ncthw_to_cl: null pointer");
    TCX_CHECK(N > 0 && N < 65536 && C > 0 && spatial > 0, TCX_E_SHAPE, "tcx_ncthw_to_cl: bad shape");
    dim3 grid((unsigned)((spatial + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL(ncthw_to_cl_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, C, spatial, mul);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_cl_to_ncthw_frames(const void* x, float* y, int32_t N, int32_t C, int64_t spatial, int64_t out_sstride,
                                      int64_t out_off, void* stream) {
    TCX_CHECK(x && y, TCX_E_NULL, "tcx_cl_to_ncthw_frames: null pointer");
    TCX_CHECK(N > 0 && N < 65536 && C > 0 && spatial > 0 && out_off >= 0 && out_off + spatial <= out_sstride, TCX_E_SHAPE,
              "tcx_cl_to_ncthw_frames: bad shape");
    dim3 grid(grid_for(spatial * C), (unsigned)N);
    hipLaunchKernelGGL(cl_to_frames_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, y, C, spatial, out_sstride, out_off);
    TCX_LAUNCH_RET();
}
