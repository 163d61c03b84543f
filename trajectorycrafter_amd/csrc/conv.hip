// bf16 implicit-GEMM convolution, channels-last, for the MAGViT/CogVideoX VAE (K11, K12, 1x1x1 convs).
//
// GEMM view:  Y^T[co, m] = sum_k W[co, k] * X[m, k],   m = (n, t, oy, ox) output position,
//             k = (dt, dy, dx, ci) with ci fastest (weights pre-permuted to [Cout, kT, kH, kW, Cin]).
// The X operand is gathered on the fly: causal temporal context from `cache` (or the first frame
// replicated), spatial zero padding, optional nearest x2 upsample (ups) and temporal frame map
// (t_map) folded into the gather index, so neither the padded, the upsampled nor the im2col tensor
// is ever materialised in HBM.
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile 128 (co) x 128 (m) x 32 (k); each wave owns
// 64 x 64 = 2 x 2 MFMA 32x32x16 tiles.  W is the MFMA A operand and X the B operand, so a lane owns
// ONE output position and 4 consecutive channels per accumulator quad -> 8-byte channels-last
// stores.  Both tiles are staged global -> registers -> LDS (issue-early / write-late, double
// buffered, one barrier per k-tile) in 64-byte rows with an XOR swizzle that makes the
// ds_read_b128 fragment reads bank-conflict free.
#include "tcx_common.h"

namespace {

struct ConvParams {
    const uint16_t *x, *cache, *w, *bias, *res;
    uint16_t* y;
    const int32_t* t_map;
    int32_t N, T_in, H_in, W_in, Cin, Cout, kT, kH, kW, T_out, H, W, ups;   // H, W: output grid
    int32_t LH, LW, stride, pad_h, pad_w;                                      // logical input grid (after the x2 upsample)
    int32_t Ktot;       // kT*kH*kW*Cin
    int64_t M;          // N*T_out*H*W
    uint32_t ntn, nwg;  // N tiles (co), total workgroups
};

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_BYTES = 128 * BK * 2;  // 8 KiB per operand tile

// 64-byte rows, 4 rows per 256-B bank row: XOR the 16-B chunk index with bits 2..3 of the row
__device__ __forceinline__ int tile_off(int row, int ch) { return row * 64 + ((ch ^ ((row >> 2) & 3)) << 4); }

__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wm = wave & 1;     // wave tile: co block wn, position block wm

    const uint32_t id = xcd_remap(blockIdx.x, p.nwg);
    const uint32_t mt = id / p.ntn, nt = id - mt * p.ntn;
    const int64_t m0 = (int64_t)mt * BM;
    const int co0 = nt * BN;

    // ---- staging assignment: each thread owns chunk (tid & 3) of rows (tid >> 2) and (tid >> 2) + 64 ----
    const int ch = tid & 3;
    const int HW = p.H * p.W;
    int xn[2], xt[2], xoy[2], xox[2];
    bool xvalid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + (tid >> 2) + 64 * i;
        xvalid[i] = m < p.M;
        const int64_t mm = xvalid[i] ? m : 0;
        const int64_t fr = mm / HW;
        const int rem = (int)(mm - fr * HW);
        xoy[i] = rem / p.W;
        xox[i] = rem - xoy[i] * p.W;
        xn[i] = (int)(fr / p.T_out);
        xt[i] = (int)(fr - (int64_t)xn[i] * p.T_out);
    }
    const int64_t frame_elems = (int64_t)p.H_in * p.W_in * p.Cin;

    u32x4 wreg[2], xreg[2];
    // k position of this thread's chunk, advanced by BK per k-tile: k = tap * Cin + ci
    int kk = 8 * ch, tap = 0, ci = 8 * ch;
    while (ci >= p.Cin) { ci -= p.Cin; ++tap; }

    auto stage_load = [&]() {
        const bool kvalid = kk < p.Ktot;
        // weights: row co, contiguous k
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = co0 + (tid >> 2) + 64 * i;
            if (kvalid && co < p.Cout) wreg[i] = *reinterpret_cast<const u32x4*>(p.w + (int64_t)co * p.Ktot + kk);
            else wreg[i] = u32x4{0, 0, 0, 0};
        }
        const int dx = tap % p.kW, t2 = tap / p.kW;
        const int dy = t2 % p.kH, dt = t2 / p.kH;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xreg[i] = u32x4{0, 0, 0, 0};
            if (!kvalid || !xvalid[i]) continue;
            const int iy = xoy[i] * p.stride + dy - p.pad_h, ix = xox[i] * p.stride + dx - p.pad_w;
            if (iy < 0 || iy >= p.LH || ix < 0 || ix >= p.LW) continue;
            const int li = xt[i] + dt;                       // logical input frame (cache frames first)
            const uint16_t* fb;
            if (li < p.kT - 1) {
                if (p.cache) fb = p.cache + ((int64_t)xn[i] * (p.kT - 1) + li) * frame_elems;
                else fb = p.x + ((int64_t)xn[i] * p.T_in + (p.t_map ? p.t_map[0] : 0)) * frame_elems;
            } else {
                const int ts = li - (p.kT - 1);
                fb = p.x + ((int64_t)xn[i] * p.T_in + (p.t_map ? p.t_map[ts] : ts)) * frame_elems;
            }
            xreg[i] = *reinterpret_cast<const u32x4*>(fb + ((int64_t)(iy >> p.ups) * p.W_in + (ix >> p.ups)) * p.Cin + ci);
        }
        kk += BK;
        ci += BK;
        while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
    };
    auto stage_write = [&](int buf) {
        char* wb = smem + buf * 2 * TILE_BYTES;
        char* xb = wb + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 2) + 64 * i;
            *reinterpret_cast<u32x4*>(wb + tile_off(row, ch)) = wreg[i];
            *reinterpret_cast<u32x4*>(xb + tile_off(row, ch)) = xreg[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int nk = (p.Ktot + BK - 1) / BK;
    stage_load();
    stage_write(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const char* wb = smem + buf * 2 * TILE_BYTES;
        const char* xb = wb + TILE_BYTES;
        const bool more = kt + 1 < nk;
        if (more) stage_load();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const bf16x8*>(wb + tile_off(64 * wn + 32 * a + r, 2 * ks + h));
#pragma unroll
            for (int b = 0; b < 2; ++b) bfr[b] = *reinterpret_cast<const bf16x8*>(xb + tile_off(64 * wm + 32 * b + r, 2 * ks + h));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
        if (more) stage_write(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns position m = m0 + 64 wm + 32 b + r and channels co0 + 64 wn + 32 a + 8 i + 4 h + (0..3) ----
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t m = m0 + 64 * wm + 32 * b + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 64 * wn + 32 * a + 8 * i + 4 * h;
                if (co >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * i + e];
                if (co + 3 < p.Cout && (p.Cout & 3) == 0) {
                    if (p.bias) {
                        const u32x2 bb = *reinterpret_cast<const u32x2*>(p.bias + co);
                        v[0] += bf16lo(bb[0]); v[1] += bf16hi(bb[0]); v[2] += bf16lo(bb[1]); v[3] += bf16hi(bb[1]);
                    }
                    if (p.res) {
                        const u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + m * p.Cout + co);
                        v[0] += bf16lo(rr[0]); v[1] += bf16hi(rr[0]); v[2] += bf16lo(rr[1]); v[3] += bf16hi(rr[1]);
                    }
                    u32x2 o;
                    o[0] = pack_bf16(v[0], v[1]);
                    o[1] = pack_bf16(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(p.y + m * p.Cout + co) = o;
                } else {
                    for (int e = 0; e < 4 && co + e < p.Cout; ++e) {
                        float t = v[e];
                        if (p.bias) t += bf16_bits_to_f32(p.bias[co + e]);
                        if (p.res) t += bf16_bits_to_f32(p.res[m * p.Cout + co + e]);
                        p.y[m * p.Cout + co + e] = (uint16_t)(pack_bf16(t, 0.f) & 0xffff);
                    }
                }
            }
    }
}

// ---- narrow-output convolution (Cout <= 4: the decoder's conv_out, 128 -> 3 at full resolution) -------------------------------
// An MFMA tile is at least 16 outputs wide (the kernels above: 128), so 3 output channels waste 81-97 % of the matrix work and,
// worse, of the X staging: 3.6 ms per 8 x 480 x 720 chunk on the kernel above.  Here 8 lanes share one output pixel: lane
// (lane & 7) owns the 16-byte channel chunk (lane & 7) of every 64-channel slice, so a wave-instruction reads 8 pixels x 128
// contiguous bytes (full lines); the products run on v_dot2c_f32_bf16 (two bf16 MACs per lane-op, fp32 accumulate) against
// the weights staged once per workgroup in LDS ([co][tap][ci], 20 KiB for 3 x 27 x 128), and the 8 partial sums of a pixel are
// folded with DPP adds.  Padding taps are skipped per pixel (they contribute exact zeros).
template <int CO>
__global__ __launch_bounds__(256) void conv_narrow_kernel(const ConvParams p, int groups_per_wave) {
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c8 = lane & 7, pl = lane >> 3;
    const int ntap = p.kT * p.kH * p.kW, slices = p.Cin >> 6;
    {   // weights -> LDS (same layout as global: [co][tap][ci])
        const int n16 = p.Cout * ntap * p.Cin / 8;
        for (int i = tid; i < n16; i += 256) reinterpret_cast<u32x4*>(wlds)[i] = reinterpret_cast<const u32x4*>(p.w)[i];
    }
    __syncthreads();
    const int HW = p.H * p.W;
    const int64_t frame_elems = (int64_t)p.H_in * p.W_in * p.Cin;
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + wave) * groups_per_wave;
    for (int gi = 0; gi < groups_per_wave; ++gi) {
        const int64_t m = (g0 + gi) * 8 + pl;                  // this lane group's output pixel
        if ((g0 + gi) * 8 >= p.M) break;                        // wave-uniform
        const bool live = m < p.M;
        const int64_t mm = live ? m : p.M - 1;
        const int64_t fr = mm / HW;
        const int rem = (int)(mm - fr * HW);
        const int oy = rem / p.W, ox = rem - oy * p.W;
        const int n = (int)(fr / p.T_out), t = (int)(fr - (int64_t)n * p.T_out);
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.f;
        for (int dt = 0; dt < p.kT; ++dt) {
            const int li = t + dt;                              // logical input frame: kT - 1 context frames first
            const uint16_t* fb;
            if (li < p.kT - 1) fb = p.cache ? p.cache + ((int64_t)n * (p.kT - 1) + li) * frame_elems : p.x + (int64_t)n * p.T_in * frame_elems;
            else fb = p.x + ((int64_t)n * p.T_in + (li - (p.kT - 1))) * frame_elems;
            const uint16_t* centre = fb + ((int64_t)(oy - p.pad_h) * p.W_in + (ox - p.pad_w)) * p.Cin + 8 * c8;
            for (int sl = 0; sl < slices; ++sl) {
                // the 3 x 3 spatial taps of this frame and 64-channel slice: nine loads in flight, then 9 x CO x 4 dot products
                u32x4 xv[9];
#pragma unroll
                for (int s9 = 0; s9 < 9; ++s9) {
                    const int dy = s9 / 3, dx = s9 % 3;
                    const bool ok = (unsigned)(oy + dy - p.pad_h) < (unsigned)p.LH && (unsigned)(ox + dx - p.pad_w) < (unsigned)p.LW;
                    xv[s9] = u32x4{0u, 0u, 0u, 0u};
                    if (ok) xv[s9] = *reinterpret_cast<const u32x4*>(centre + ((int64_t)dy * p.W_in + dx) * p.Cin + 64 * sl);
                }
#pragma unroll
                for (int s9 = 0; s9 < 9; ++s9) {
                    const int tap = dt * 9 + s9;
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        if (co < p.Cout) {
                            const u32x4 wv = *reinterpret_cast<const u32x4*>(wlds + (((size_t)co * ntap + tap) * p.Cin + 64 * sl + 8 * c8) * 2);
                            // inline asm: with __builtin_amdgcn_fdot2_f32_bf16 on the elements of two u32x4 values hipcc (ROCm 7.2)
                            // emitted all four products on element 0 (and loaded only that dword)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t xa = xv[s9][j], wa = wv[j];
                                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[co]) : "v"(xa), "v"(wa));
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = group8_sum(acc[co]);
        if (live && c8 == 0) {
#pragma unroll
            for (int co = 0; co < CO; ++co)
                if (co < p.Cout) {
                    const float v = acc[co] + (p.bias ? bf16_bits_to_f32(p.bias[co]) : 0.f);
                    p.y[m * p.Cout + co] = (uint16_t)(pack_bf16(v, 0.f) & 0xffff);
                }
        }
    }
}

}  // namespace

// TCX_CONV_GENERIC=1 (read once): run every conv on this file's register-staged kernel — A/B timing and parity cross-checks
// of the two kernels; never set in production.
#include <stdlib.h>
static bool tcx_conv_force_generic() {
    static const bool v = [] { const char* e = getenv("TCX_CONV_GENERIC"); return e && e[0] == '1'; }();
    return v;
}

// The ONE dispatch decision (tcx_conv3d_cl launches what this returns; tcx_conv3d_route reports it).
static int conv_route(const TcxConvArgs& a) {
    if (!tcx_conv_force_generic()) {
        if (tcx_conv_mfma_supported(a)) return a.Cout >= 256 ? TCX_CONV_ROUTE_MFMA_WIDE : TCX_CONV_ROUTE_MFMA_TALL;
        if (a.Cout <= 4 && a.Cin % 64 == 0 && a.stride == 1 && a.ups == 0 && !a.t_map && !a.res && a.kH == 3 && a.kW == 3 &&
            a.H_out == a.H_in && a.W_out == a.W_in && (int64_t)a.Cout * a.kT * a.kH * a.kW * a.Cin * 2 <= 64 * 1024)
            return TCX_CONV_ROUTE_NARROW;
    }
    return TCX_CONV_ROUTE_IGEMM;
}

extern "C" int tcx_conv3d_route(int32_t T_in, int32_t H_in, int32_t W_in, int32_t Cin, int32_t Cout, int32_t kT, int32_t kH,
                                int32_t kW, int32_t ups, int32_t stride, int32_t H_out, int32_t W_out, int32_t has_t_map,
                                int32_t has_res) {
    static const int32_t dummy = 0;
    TcxConvArgs a{};
    a.t_map = has_t_map ? &dummy : nullptr;
    a.res = has_res ? &dummy : nullptr;
    a.N = 1; a.T_in = T_in; a.H_in = H_in; a.W_in = W_in; a.Cin = Cin; a.Cout = Cout; a.kT = kT; a.kH = kH; a.kW = kW;
    a.T_out = T_in; a.ups = ups; a.stride = stride; a.H_out = H_out; a.W_out = W_out;
    return conv_route(a);
}

extern "C" int tcx_conv3d_cl(const void* x, const void* cache, const void* w, const void* bias, const void* res, void* y,
                             int32_t N, int32_t T_in, int32_t H_in, int32_t W_in, int32_t Cin, int32_t Cout,
                             int32_t kT, int32_t kH, int32_t kW, int32_t T_out, int32_t ups, int32_t stride,
                             int32_t pad_h, int32_t pad_w, int32_t H_out, int32_t W_out, const int32_t* t_map,
                             void* stream) {
    TCX_CHECK(x && w && y, TCX_E_NULL, "tcx_conv3d_cl: null x/w/y");
    TCX_CHECK(N > 0 && T_in > 0 && H_in > 0 && W_in > 0 && Cin > 0 && Cout > 0 && T_out > 0, TCX_E_SHAPE, "tcx_conv3d_cl: empty shape");
    TCX_CHECK(Cin % 8 == 0, TCX_E_SHAPE, "tcx_conv3d_cl: Cin (%d) must be a multiple of 8 (pad channels on the host)", Cin);
    TCX_CHECK(kT >= 1 && kT <= 3 && (kH == 1 || kH == 3) && (kW == 1 || kW == 3), TCX_E_SHAPE, "tcx_conv3d_cl: kernel %dx%dx%d unsupported", kT, kH, kW);
    TCX_CHECK(ups == 0 || ups == 1, TCX_E_SHAPE, "tcx_conv3d_cl: ups must be 0 or 1");
    TCX_CHECK(stride == 1 || stride == 2, TCX_E_SHAPE, "tcx_conv3d_cl: stride must be 1 or 2");
    TCX_CHECK(pad_h >= 0 && pad_h < kH && pad_w >= 0 && pad_w < kW && H_out > 0 && W_out > 0, TCX_E_SHAPE, "tcx_conv3d_cl: bad padding / output size");
    TCX_CHECK((H_out - 1) * stride - pad_h < (H_in << ups) && (W_out - 1) * stride - pad_w < (W_in << ups), TCX_E_SHAPE,
              "tcx_conv3d_cl: output grid %dx%d reads past the input", H_out, W_out);
    TCX_CHECK(t_map != nullptr || T_out == T_in, TCX_E_SHAPE, "tcx_conv3d_cl: T_out (%d) != T_in (%d) needs a t_map", T_out, T_in);
    TCX_CHECK(!(kT > 1 && t_map), TCX_E_SHAPE, "tcx_conv3d_cl: t_map is only supported with kT == 1");
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(cache) && tcx_aligned16(w) && tcx_aligned16(y) && tcx_aligned16(res) &&
                  (reinterpret_cast<uintptr_t>(bias) & 7) == 0,
              TCX_E_ALIGN, "tcx_conv3d_cl: pointers must be 16-byte aligned (bias 8)");
    const TcxConvArgs a{x, cache, w, bias, res, y, t_map, N, T_in, H_in, W_in, Cin, Cout, kT, kH, kW, T_out, ups, stride, pad_h, pad_w, H_out, W_out};
    const int route = conv_route(a);
    if (route == TCX_CONV_ROUTE_MFMA_WIDE || route == TCX_CONV_ROUTE_MFMA_TALL)
        return tcx_conv_mfma_launch(a, (hipStream_t)stream);                                   // LDS-DMA GEMM pipeline (conv_mfma.hip)
    const bool narrow = route == TCX_CONV_ROUTE_NARROW;
    ConvParams p;
    p.x = (const uint16_t*)x; p.cache = (const uint16_t*)cache; p.w = (const uint16_t*)w; p.bias = (const uint16_t*)bias;
    p.res = (const uint16_t*)res; p.y = (uint16_t*)y; p.t_map = t_map;
    p.N = N; p.T_in = T_in; p.H_in = H_in; p.W_in = W_in; p.Cin = Cin; p.Cout = Cout; p.kT = kT; p.kH = kH; p.kW = kW;
    p.T_out = T_out; p.H = H_out; p.W = W_out; p.ups = ups;
    p.LH = H_in << ups; p.LW = W_in << ups; p.stride = stride; p.pad_h = pad_h; p.pad_w = pad_w;
    p.Ktot = kT * kH * kW * Cin;
    p.M = (int64_t)N * T_out * p.H * p.W;
    p.ntn = (uint32_t)((Cout + BN - 1) / BN);
    const int64_t nwg = ((p.M + BM - 1) / BM) * p.ntn;
    TCX_CHECK(nwg < (1ll << 31), TCX_E_SHAPE, "tcx_conv3d_cl: grid too large");
    p.nwg = (uint32_t)nwg;
    if (narrow) {                                            // Cout <= 4: dot-product kernel, weights in LDS
        const int gpw = 16;                                  // 8-pixel groups per wave: 512 pixels per workgroup
        const int64_t nblk = (p.M + 4 * 8 * gpw - 1) / (4 * 8 * gpw);
        TCX_CHECK(nblk < (1ll << 31), TCX_E_SHAPE, "tcx_conv3d_cl: grid too large");
        hipLaunchKernelGGL(conv_narrow_kernel<4>, dim3((unsigned)nblk), dim3(256), (size_t)Cout * kT * kH * kW * Cin * 2, (hipStream_t)stream, p, gpw);
        TCX_LAUNCH_RET();
    }
    hipLaunchKernelGGL(conv_igemm_kernel, dim3(p.nwg), dim3(256), 0, (hipStream_t)stream, p);
    TCX_LAUNCH_RET();
}
