// Point-cloud render: forward warp of a frame into a new view by bilinear splatting.
// Replaces the 49 sequential torch splats of Warper.forward_warp (reference models/utils.py:220-293, 350-583; 8
// index_put_(accumulate=True) each) with three HBM/atomic-bound kernels, fp32 like the reference:
//   project : per source pixel  K2 [R|t] (depth K1^-1 (x,y,1)) -> flow, target depth; max of log(1+depth) by atomic max
//   splat   : per source pixel 4 corners x 5 floats (r, g, b, depth, weight) of float atomic adds into a padded
//             (h+2) x (w+2) accumulator (one 20-byte record per target pixel: the five adds of a corner share a line)
//   resolve : normalise, clamp, mask.
// Float atomics commute only up to rounding: results match the oracle to ~1e-5, not bitwise (the reference's own
// index_put_ accumulate on a GPU has the same property).
#include "tcx_common.h"
#include <algorithm>

namespace {

struct WarpParams {
    const float *frame, *mask1, *depth, *mats;
    float *flow, *tdepth, *acc, *warped, *mask2, *wdepth;
    unsigned* logmax;
    int32_t b, h, w, per_item, clean;
};

// mats[n] = { K1inv (9, row major), Rel (12: 3 x 4 rows of [R|t]), K2 (9) }
// grid = (blocks per item, b): a block stays inside one item, so the max of log(1+depth) costs one atomic per block.
__global__ __launch_bounds__(256) void warp_project_kernel(const WarpParams p) {
    const int hw = p.h * p.w, n = blockIdx.y;
    const float* m = p.mats + 30 * n;
    float k[30];
#pragma unroll
    for (int j = 0; j < 30; ++j) k[j] = m[j];
    float lmax = 0.f;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < hw; pix += gridDim.x * blockDim.x) {
        const int x = pix % p.w, y = pix / p.w;
        const int64_t i = (int64_t)n * hw + pix;
        const float fx = (float)x, fy = (float)y, d = p.depth[i];
        float ray[3], cam[3], pr[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) ray[r] = (k[3 * r] * fx + k[3 * r + 1] * fy + k[3 * r + 2]) * d;
#pragma unroll
        for (int r = 0; r < 3; ++r) cam[r] = (k[9 + 4 * r] * ray[0] + k[9 + 4 * r + 1] * ray[1] + k[9 + 4 * r + 2] * ray[2]) + k[9 + 4 * r + 3];
#pragma unroll
        for (int r = 0; r < 3; ++r) pr[r] = k[21 + 3 * r] * cam[0] + k[21 + 3 * r + 1] * cam[1] + k[21 + 3 * r + 2] * cam[2];
        if (cam[2] <= 0.01f) pr[0] = pr[1] = pr[2] = 1000.0f;                  // behind the target camera (:403-417)
        p.flow[(2 * (int64_t)n) * hw + pix] = pr[0] / pr[2] - fx;
        p.flow[(2 * (int64_t)n + 1) * hw + pix] = pr[1] / pr[2] - fy;
        p.tdepth[i] = pr[2];
        lmax = fmaxf(lmax, logf(1.0f + fminf(fmaxf(pr[2], 0.f), 1000.0f)));
    }
    // max of log(1+depth) over the batch (reference :478-479) or per item; non-negative floats order like their bits
    __shared__ float smax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = lmax;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(p.logmax + (p.per_item ? n : 0), __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))));
}

// Splat.  Global float atomics are the bound (one 4-byte add per lane-op at the memory side), and the four corners of
// neighbouring source pixels land on the same target pixels — so a workgroup first bins its 32x32 source tile into an
// LDS window (ds_add_f32) placed around where the tile's centre lands, then flushes the touched window pixels with one
// global atomic per float: ~4x fewer global atomics for near-unit magnification.  Corners that fall outside the
// window (strong magnification, depth edges, points pushed behind the camera) go straight to global memory.
constexpr int kTile = 32, kWinX = 48, kWinY = 44;

__global__ __launch_bounds__(256) void warp_splat_kernel(const WarpParams p) {
    __shared__ float win[5][kWinY][kWinX];
    __shared__ int org[2];
    const int hw = p.h * p.w, n = blockIdx.z;
    const int x0 = blockIdx.x * kTile, y0 = blockIdx.y * kTile;
    const int W2 = p.w + 2, H2 = p.h + 2;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float lim_x = (float)(p.w + 1), lim_y = (float)(p.h + 1);
    const float logmax = __uint_as_float(p.logmax[p.per_item ? n : 0]);
    for (int j = threadIdx.x; j < 5 * kWinY * kWinX; j += 256) (&win[0][0][0])[j] = 0.f;

    float wgt[4][4], val[4][4];
    int cix[4][2], ciy[4][2];
    bool valid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        valid[j] = x0 + tx < p.w && y0 + ty + 8 * j < p.h;
        const int x = min(x0 + tx, p.w - 1), y = min(y0 + ty + 8 * j, p.h - 1);   // ragged tiles: stand-ins, not splatted
        const int pix = y * p.w + x;
        const int64_t i = (int64_t)n * hw + pix;
        const float td = p.tdepth[i];
        float px = p.flow[(2 * (int64_t)n) * hw + pix] + (float)x + 1.0f;       // +1: border of the padded accumulator
        float py = p.flow[(2 * (int64_t)n + 1) * hw + pix] + (float)y + 1.0f;
        // floor / ceil BEFORE clamping, each then clamped on its own (reference :455-476)
        const float flx = fminf(fmaxf(floorf(px), 0.f), lim_x), fly = fminf(fmaxf(floorf(py), 0.f), lim_y);
        const float cex = fminf(fmaxf(ceilf(px), 0.f), lim_x), cey = fminf(fmaxf(ceilf(py), 0.f), lim_y);
        px = fminf(fmaxf(px, 0.f), lim_x);
        py = fminf(fmaxf(py, 0.f), lim_y);
        const float dfx = px - flx, dfy = py - fly, dcx = cex - px, dcy = cey - py;
        const float dw = expf(logf(1.0f + fminf(fmaxf(td, 0.f), 1000.0f)) / logmax * 50.0f);
        const float base = (p.mask1 ? p.mask1[i] : 1.0f) / dw;
        val[j][0] = p.frame[(3 * (int64_t)n) * hw + pix];
        val[j][1] = p.frame[(3 * (int64_t)n + 1) * hw + pix];
        val[j][2] = p.frame[(3 * (int64_t)n + 2) * hw + pix];
        val[j][3] = td;
        cix[j][0] = (int)flx; cix[j][1] = (int)cex;
        ciy[j][0] = (int)fly; ciy[j][1] = (int)cey;
        const float wx[2] = {1.0f - dfx, 1.0f - dcx}, wy[2] = {1.0f - dfy, 1.0f - dcy};
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) wgt[j][2 * cy + cx] = wy[cy] * wx[cx] * base;
        // window origin: where the tile's centre pixel lands, minus half a window
        if (j == 2 && tx == 16 && ty == 0) {             // pixel (x0+16, y0+16), or the tile's last valid one
            org[0] = cix[j][0] - kWinX / 2;
            org[1] = ciy[j][0] - kWinY / 2;
        }
    }
    __syncthreads();
    const int ox = org[0], oy = org[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!valid[j]) continue;
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                const float wg = wgt[j][2 * cy + cx];
                const int gx = cix[j][cx], gy = ciy[j][cy];
                const int lx = gx - ox, ly = gy - oy;
                if ((unsigned)lx < (unsigned)kWinX && (unsigned)ly < (unsigned)kWinY) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) atomicAdd(&win[c][ly][lx], val[j][c] * wg);
                    atomicAdd(&win[4][ly][lx], wg);
                } else {
                    float* dst = p.acc + (((int64_t)n * H2 + gy) * W2 + gx) * 5;
#pragma unroll
                    for (int c = 0; c < 4; ++c) atomicAdd(dst + c, val[j][c] * wg);
                    atomicAdd(dst + 4, wg);
                }
            }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < kWinY * kWinX; j += 256) {
        const int ly = j / kWinX, lx = j - ly * kWinX;
        const float wsum = win[4][ly][lx];
        if (wsum != 0.f) {                               // only in-range corners were binned: origin + (lx, ly) is inside the accumulator
            float* dst = p.acc + (((int64_t)n * H2 + (oy + ly)) * W2 + (ox + lx)) * 5;
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(dst + c, win[c][ly][lx]);
            atomicAdd(dst + 4, wsum);
        }
    }
}

__global__ __launch_bounds__(256) void warp_resolve_kernel(const WarpParams p) {
    const int64_t hw = (int64_t)p.h * p.w, total = (int64_t)p.b * hw;
    const int W2 = p.w + 2, H2 = p.h + 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.w), y = (int)((i / p.w) % p.h), n = (int)(i / hw);
        const int64_t pix = i - (int64_t)n * hw;
        const float* a = p.acc + (((int64_t)n * H2 + y + 1) * W2 + x + 1) * 5;
        const float wsum = a[4];
        const bool hit = wsum > 0.f;
        bool keep = hit;
        if (p.clean) {                                   // clean_points (:585-626): holes dilated by a 5x5 box
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w)
                        keep = keep && (a[((int64_t)dy * W2 + dx) * 5 + 4] > 0.f);
                }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float o = hit ? a[c] / wsum : -1.0f;
            o = fminf(fmaxf(o, -1.0f), 1.0f);
            if (p.clean) o = ((o + 1.0f) / 2.0f) * (keep ? 1.0f : 0.0f) * 2.0f - 1.0f;
            p.warped[(3 * (int64_t)n + c) * hw + pix] = o;
        }
        p.wdepth[i] = hit ? a[3] / wsum : 0.0f;          // the depth map is not cleaned (:287-293)
        p.mask2[i] = keep ? 1.0f : 0.0f;
    }
}

// ---- generic bilinear splat: Warper.bilinear_splatting (reference models/utils.py:422-583) of an arbitrary C <= 4 channel image
// along a GIVEN flow with depth weights from a GIVEN depth map — the building block of forward_warp(twice=True) (:294-347), which is
// off the inference path: plain global atomics, no LDS window.
struct SplatParams {
    const float *src, *mask1, *depth, *flow;
    float *acc, *out, *mask2;
    unsigned* logmax;
    int32_t b, c, h, w, is_image;
    float flow_scale;
};

__global__ __launch_bounds__(256) void splat_logmax_kernel(const SplatParams p) {
    const int64_t total = (int64_t)p.b * p.h * p.w;
    float lmax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        lmax = fmaxf(lmax, logf(1.0f + fminf(fmaxf(p.depth[i], 0.f), 1000.0f)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(p.logmax, __float_as_uint(lmax));
}

__global__ __launch_bounds__(256) void splat_generic_kernel(const SplatParams p) {
    const int64_t hw = (int64_t)p.h * p.w, total = (int64_t)p.b * hw;
    const int W2 = p.w + 2, H2 = p.h + 2;
    const float lim_x = (float)(p.w + 1), lim_y = (float)(p.h + 1);
    const float logmax = __uint_as_float(*p.logmax);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / hw);
        const int64_t pix = i - (int64_t)n * hw;
        const int x = (int)(pix % p.w), y = (int)(pix / p.w);
        float px = p.flow[(2 * (int64_t)n) * hw + pix] * p.flow_scale + (float)x + 1.0f;
        float py = p.flow[(2 * (int64_t)n + 1) * hw + pix] * p.flow_scale + (float)y + 1.0f;
        const float flx = fminf(fmaxf(floorf(px), 0.f), lim_x), fly = fminf(fmaxf(floorf(py), 0.f), lim_y);
        const float cex = fminf(fmaxf(ceilf(px), 0.f), lim_x), cey = fminf(fmaxf(ceilf(py), 0.f), lim_y);
        px = fminf(fmaxf(px, 0.f), lim_x);
        py = fminf(fmaxf(py, 0.f), lim_y);
        const float wx[2] = {1.0f - (px - flx), 1.0f - (cex - px)}, wy[2] = {1.0f - (py - fly), 1.0f - (cey - py)};
        const int ix[2] = {(int)flx, (int)cex}, iy[2] = {(int)fly, (int)cey};
        const float dw = expf(logf(1.0f + fminf(fmaxf(p.depth[i], 0.f), 1000.0f)) / logmax * 50.0f);
        const float base = (p.mask1 ? p.mask1[i] : 1.0f) / dw;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < p.c; ++ch) v[ch] = p.src[((int64_t)n * p.c + ch) * hw + pix];
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                const float wg = wy[cy] * wx[cx] * base;
                float* dst = p.acc + (((int64_t)n * H2 + iy[cy]) * W2 + ix[cx]) * 5;
                for (int ch = 0; ch < p.c; ++ch) atomicAdd(dst + ch, v[ch] * wg);
                atomicAdd(dst + 4, wg);
            }
    }
}

__global__ __launch_bounds__(256) void splat_resolve_kernel(const SplatParams p) {
    const int64_t hw = (int64_t)p.h * p.w, total = (int64_t)p.b * hw;
    const int W2 = p.w + 2, H2 = p.h + 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.w), y = (int)((i / p.w) % p.h), n = (int)(i / hw);
        const int64_t pix = i - (int64_t)n * hw;
        const float* a = p.acc + (((int64_t)n * H2 + y + 1) * W2 + x + 1) * 5;
        const float wsum = a[4];
        const bool hit = wsum > 0.f;
        for (int ch = 0; ch < p.c; ++ch) {
            float o = hit ? a[ch] / wsum : (p.is_image ? -1.0f : 0.0f);
            if (p.is_image) o = fminf(fmaxf(o, -1.0f), 1.0f);
            p.out[((int64_t)n * p.c + ch) * hw + pix] = o;
        }
        p.mask2[i] = hit ? 1.0f : 0.0f;
    }
}

inline unsigned wgrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b > 256 * 8 ? 256 * 8 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int tcx_warp_forward(const float* frame, const float* mask1, const float* depth, const float* mats,
                                float* flow, float* tdepth, float* acc, float* warped, float* mask2, float* wdepth,
                                int32_t b, int32_t h, int32_t w, int32_t flags, void* stream) {
    TCX_CHECK(frame && depth && mats && flow && tdepth && acc && warped && mask2 && wdepth, TCX_E_NULL, "tcx_warp_forward: null pointer");
    TCX_CHECK(b > 0 && h > 0 && w > 0, TCX_E_SHAPE, "tcx_warp_forward: empty shape");
    hipStream_t st = (hipStream_t)stream;
    TCX_CHECK((flags & ~(TCX_WARP_PER_ITEM_MAX | TCX_WARP_CLEAN_POINTS)) == 0, TCX_E_SHAPE, "tcx_warp_forward: unknown flags 0x%x", flags);
    const size_t acc_bytes = sizeof(float) * ((size_t)b * (h + 2) * (w + 2) * 5 + b);     // + b words: max log-depth
    hipError_t e = hipMemsetAsync(acc, 0, acc_bytes, st);
    if (e != hipSuccess) { tcx_set_error("tcx_warp_forward: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    WarpParams p{frame, mask1, depth, mats, flow, tdepth, acc, warped, mask2, wdepth,
                 reinterpret_cast<unsigned*>(acc + (size_t)b * (h + 2) * (w + 2) * 5), b, h, w,
                 (flags & TCX_WARP_PER_ITEM_MAX) ? 1 : 0, (flags & TCX_WARP_CLEAN_POINTS) ? 1 : 0};
    const int64_t total = (int64_t)b * h * w;
    TCX_CHECK(b <= 65535 && (h + kTile - 1) / kTile <= 65535, TCX_E_SHAPE, "tcx_warp_forward: b=%d / h=%d exceed the grid limits", b, h);
    const int64_t hw = (int64_t)h * w;
    TCX_CHECK(hw < (1ll << 30), TCX_E_SHAPE, "tcx_warp_forward: h*w = %lld too large", (long long)hw);
    const unsigned pblocks = (unsigned)std::min<int64_t>((hw + 255) / 256, std::max<int64_t>(1, 4096 / b));
    hipLaunchKernelGGL(warp_project_kernel, dim3(pblocks, b), dim3(256), 0, st, p);
    hipLaunchKernelGGL(warp_splat_kernel, dim3((w + kTile - 1) / kTile, (h + kTile - 1) / kTile, b), dim3(256), 0, st, p);
    hipLaunchKernelGGL(warp_resolve_kernel, dim3(wgrid(total)), dim3(256), 0, st, p);
    TCX_LAUNCH_RET();
}

extern "C" int tcx_bilinear_splat(const float* src, const float* mask1, const float* depth, const float* flow, float* acc, float* out,
                                  float* mask2, int32_t b, int32_t c, int32_t h, int32_t w, int32_t is_image, float flow_scale,
                                  void* stream) {
    TCX_CHECK(src && depth && flow && acc && out && mask2, TCX_E_NULL, "tcx_bilinear_splat: null pointer");
    TCX_CHECK(b > 0 && h > 0 && w > 0 && c >= 1 && c <= 4, TCX_E_SHAPE, "tcx_bilinear_splat: need 1 <= channels <= 4 and a non-empty image");
    TCX_CHECK((int64_t)b * h * w < (1ll << 40), TCX_E_SHAPE, "tcx_bilinear_splat: image too large");
    hipStream_t st = (hipStream_t)stream;
    const size_t acc_floats = (size_t)b * (h + 2) * (w + 2) * 5 + 1;                       // + 1 word: max log-depth
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(float) * acc_floats, st);
    if (e != hipSuccess) { tcx_set_error("tcx_bilinear_splat: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    SplatParams p{src, mask1, depth, flow, acc, out, mask2, reinterpret_cast<unsigned*>(acc + acc_floats - 1), b, c, h, w, is_image ? 1 : 0, flow_scale};
    const int64_t total = (int64_t)b * h * w;
    hipLaunchKernelGGL(splat_logmax_kernel, dim3(wgrid(total)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(splat_generic_kernel, dim3(wgrid(total)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(splat_resolve_kernel, dim3(wgrid(total)), dim3(256), 0, st, p);
    TCX_LAUNCH_RET();
}
