// Issue-cost microbenchmark (gfx950): cycles per instruction for independent streams of one instruction kind,
// one or two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/exp/issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int KIND>
__global__ __launch_bounds__(512) void k(long long* out, float seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < 256; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(double*)&a[i]));
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
        } else if constexpr (KIND == 7) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
        } else if constexpr (KIND == 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (KIND == 9) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (KIND == 10) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        } else if constexpr (KIND == 11) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&a[i]));
        } else if constexpr (KIND == 12) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_ldexp_f32 %0, %0, 3" : "+v"(a[i]));
        } else if constexpr (KIND == 13) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(fa[0]), "v"(fb[0]));
        } else if constexpr (KIND == 14) {      // group with one dot2 instead of two adds
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pk_bf16_f32 %4, %4, %4\n\tv_dot2_f32_bf16 %2, %4, %3, %2"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]));
            }
        } else if constexpr (KIND == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
        } else if constexpr (KIND == 5) {      // 1 MFMA + 2 exp + 2 add + 1 cvt, x4
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_cvt_pk_bf16_f32 %4, %4, %4"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]));
            }
        } else if constexpr (KIND == 6) {      // half the waves MFMA only, half VALU only (ping-pong content)
            if (threadIdx.x < 256) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_cvt_pk_bf16_f32 %4, %4, %4"
                                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]));
            }
        }
    }
    const long long t1 = clock64();
    float sink = 0.f;
    for (int i = 0; i < 8; ++i) sink += a[i];
    for (int j = 0; j < 4; ++j) sink += acc[j][0];
    if (sink == 12345.678f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND>
void run(const char* name, int threads, int per_iter) {
    long long* d; hipMalloc(&d, 4096 * 8); hipMemset(d, 0, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %3d thr: wave0 %6.2f cyc/instr, last wave %6.2f  (kernel %.1f us => %.2f GHz if counts are cycles)\n", name, threads,
           (double)h[0] / (256.0 * per_iter), (double)h[threads / 64 - 1] / (256.0 * per_iter), ms * 1e3, h[0] / (ms * 1e6));
    hipFree(d);
}
int main() {
    for (int threads : {256, 512}) {
        run<0>("v_exp_f32 x8", threads, 8);
        run<1>("v_add_f32 x8", threads, 8);
        run<2>("v_pk_add_f32 x4", threads, 4);
        run<3>("v_cvt_pk_bf16_f32 x8", threads, 8);
        run<7>("v_exp_f16 x8", threads, 8);
        run<8>("v_exp_legacy_f32 x8", threads, 8);
        run<9>("v_rcp_f32 x8", threads, 8);
        run<10>("v_fma_f32 x8", threads, 8);
        run<11>("v_pk_fma_f32 x4", threads, 4);
        run<12>("v_ldexp_f32 x8", threads, 8);
        run<13>("v_dot2_f32_bf16 x8", threads, 8);
        run<14>("4 x {mfma, 2 exp, cvt, dot2} = group'", threads, 4);
        run<4>("mfma 32x32x16 bf16 x4 (indep)", threads, 4);
        run<5>("4 x {mfma, 2 exp, 2 add, cvt} per 'instr' = group", threads, 4);
    }
    run<6>("ping-pong content: waves 0-3 mfma, 4-7 valu group", 512, 4);
    return 0;
}
