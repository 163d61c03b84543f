// Shared device/host helpers for libtcx_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tcx_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define TCX_WAVE 64

// ---- host-side error plumbing (defined in tcx_api.cpp) ----
void tcx_set_error(const char* fmt, ...);
#define TCX_CHECK(cond, code, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            tcx_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)
#define TCX_LAUNCH_RET()                                            \
    do {                                                            \
        hipError_t e__ = hipGetLastError();                         \
        if (e__ != hipSuccess) {                                    \
            tcx_set_error("launch failed: %s", hipGetErrorString(e__)); \
            return (int)e__;                                        \
        }                                                           \
        return TCX_OK;                                              \
    } while (0)

static inline bool tcx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- device helpers ----
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// round-to-nearest-even via the hardware convert (keeps NaN a NaN, see MI355X_MICROARCH.md)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16lo(v[i]);
        f[2 * i + 1] = bf16hi(v[i]);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Bijective XCD-aware block remap: blocks that share `orig % 8` run on one XCD (observed
// round-robin dispatch); give each XCD a contiguous range of logical ids.  Speed only.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t orig, uint32_t nwg) {
    const uint32_t xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}
