// Which bf16 MFMA shape carries the attention loop's instruction group better on gfx950?  (round 3)
// Per 32 x 32 x 16 MFMA the bound-centred D = 64 loop issues {2 v_exp_f32, 2 v_add_f32, 1 v_cvt_pk_bf16_f32}.  The same FLOPs as
// two 16 x 16 x 32 MFMAs.  This microbenchmark runs both groups (and the bare MFMA streams) on RANDOM operands, one or two waves per
// SIMD, and reports per group: shader cycles (s_memtime), the in-kernel clock (s_memtime / s_memrealtime x 100 MHz, MI355X_MICROARCH.md
// DVFS item 6) and wall time, after ~1 s of back-to-back launches of the same kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_shape_group.hip -o /tmp/mfma_shape_group && /tmp/mfma_shape_group
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

constexpr int ITERS = 4096;
// KIND 0: 4 x {mfma32, 2 exp, 2 add, cvt}     1: 4 x {2 mfma16, 2 exp, 2 add, cvt}     2: 4 x {mfma16, exp, add | mfma16, exp, add, cvt}
//      3: 4 x mfma32                          4: 8 x mfma16                            5: the VALU part alone
template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long* out, uint32_t seed) {
    const uint32_t t = hash(seed ^ (blockIdx.x * 512u + threadIdx.x));
    bf16x8 fa[2], fb[2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
            fa[j][i] = (__bf16)(((int)(hash(t + 17 * i + 3 * j) & 0xffff) - 32768) * (1.0f / 32768.0f));
            fb[j][i] = (__bf16)(((int)(hash(t + 131 * i + 7 * j + 1) & 0xffff) - 32768) * (1.0f / 32768.0f));
        }
    float a[5];
    for (int i = 0; i < 5; ++i) a[i] = -1.0f - (hash(t + i) & 1023) * (1.0f / 256.0f);
    f32x16 acc[4];
    f32x4 acc4[8];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc4[j][i] = 0.f;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < ITERS; ++it) {
        float e0, e1;
#define VALU_GROUP asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %4, %0, %1" \
                                : "=&v"(e0), "=&v"(e1), "+v"(a[2]), "+v"(a[3]), "=v"(a[4]) : "v"(a[0]), "v"(a[1]));
#define VALU_HALF_A asm volatile("v_exp_f32 %0, %2\n\tv_add_f32 %1, %1, %0" : "=&v"(e0), "+v"(a[2]) : "v"(a[0]));
#define VALU_HALF_B asm volatile("v_exp_f32 %0, %3\n\tv_add_f32 %1, %1, %0\n\tv_cvt_pk_bf16_f32 %2, %4, %0" : "=&v"(e1), "+v"(a[3]), "=v"(a[4]) : "v"(a[1]), "v"(e0));
        if constexpr (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j & 1], fb[j & 1], acc[j], 0, 0, 0);
                VALU_GROUP
            }
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc4[2 * j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j & 1], fb[j & 1], acc4[2 * j], 0, 0, 0);
                acc4[2 * j + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1 - (j & 1)], fb[j & 1], acc4[2 * j + 1], 0, 0, 0);
                VALU_GROUP
            }
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc4[2 * j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j & 1], fb[j & 1], acc4[2 * j], 0, 0, 0);
                VALU_HALF_A
                acc4[2 * j + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1 - (j & 1)], fb[j & 1], acc4[2 * j + 1], 0, 0, 0);
                VALU_HALF_B
            }
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j & 1], fb[j & 1], acc[j], 0, 0, 0);
        } else if constexpr (KIND == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j & 1], fb[(j >> 1) & 1], acc4[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { VALU_GROUP }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    for (int i = 0; i < 5; ++i) sink += a[i];
    for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][7];
    for (int j = 0; j < 8; ++j) sink += acc4[j][0] + acc4[j][3];
    if (sink == 12345.678f) out[100000] = 1;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = c1 - c0;
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}
template <int KIND>
void run(const char* name, int threads) {
    unsigned long long* d; hipMalloc(&d, 256 * 8 * 2 * 8 + 1024 * 1024); hipMemset(d, 0, 256 * 8 * 2 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // ~1 s of back-to-back launches so that the clock settles under this kernel, then time 10 launches
    hipEventRecord(e0);
    float ms = 0.f; int warm = 0;
    while (ms < 1000.f) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, 1234u + warm + i);
        warm += 20;
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, 99u + i);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 8 * 2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double cyc_first = 0, cyc_last = 0, clk = 0; int n = 0;
    for (int b = 0; b < 256; ++b) {
        cyc_first += (double)h[(b * 8 + 0) * 2];
        cyc_last += (double)h[(b * 8 + nw - 1) * 2];
        for (int w = 0; w < nw; ++w) { clk += (double)h[(b * 8 + w) * 2] / (double)h[(b * 8 + w) * 2 + 1] * 0.1; ++n; }
    }
    const double groups = (double)ITERS * 4;
    const double flop_per_group = KIND == 5 ? 0.0 : 2.0 * 32 * 32 * 16 * 1.0;          // one mfma32 or two mfma16 per group
    const double tflops = flop_per_group * groups * (threads / 64) * 256 / (ms / 10 * 1e-3) / 1e12;
    printf("%-58s %d waves/SIMD: %7.2f cyc/group (wave 0) %7.2f (last wave)  clock %.3f GHz  kernel %8.1f us  %7.1f TFLOP/s\n", name, threads / 256,
           cyc_first / 256 / groups, cyc_last / 256 / groups, clk / n, ms / 10 * 1e3, tflops);
    hipFree(d);
}
int main() {
    for (int threads : {256, 512}) {
        run<3>("bare mfma 32x32x16 (1 per group)", threads);
        run<4>("bare mfma 16x16x32 (2 per group)", threads);
        run<5>("VALU part alone {2 exp, 2 add, cvt}", threads);
        run<0>("{mfma32, 2 exp, 2 add, cvt}", threads);
        run<1>("{2 mfma16, 2 exp, 2 add, cvt}", threads);
        run<2>("{mfma16, exp, add | mfma16, exp, add, cvt}", threads);
    }
    return 0;
}
