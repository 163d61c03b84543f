// Shared device/host helpers for libtcx_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property of a kernel: a process that drives several GPUs must
// set it on each of them.  One of these per kernel instantiation: a bit per device ordinal (ordinals >= 64 simply set the
// attribute on every launch), written with an atomic OR so that racing host threads at worst set the attribute twice.
struct TcxPerDeviceOnce {
    std::atomic<uint64_t> done{0};
};
static inline int tcx_ensure_dynamic_lds(TcxPerDeviceOnce& once, const void* func, int bytes, const char* what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { tcx_set_error("%s: hipGetDevice: %s", what, hipGetErrorString(e)); return (int)e; }
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && ((once.done.load(std::memory_order_acquire) >> dev) & 1ull)) return TCX_OK;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        tcx_set_error("%s: cannot reserve %d bytes of LDS on device %d: %s", what, bytes, dev, hipGetErrorString(e));
        return (int)e;
    }
    if (tracked) once.done.fetch_or(1ull << dev, std::memory_order_release);
    return TCX_OK;
}

// number of CUs of the current device, cached per device ordinal (tcx_api.cpp); 0 if it cannot be queried
uint32_t tcx_cu_count();

// ---- conv dispatch (conv.hip -> conv_mfma.hip) ----
struct TcxConvArgs {
    const void *x, *cache, *w, *bias, *res;
    void* y;
    const int32_t* t_map;
    int32_t N, T_in, H_in, W_in, Cin, Cout, kT, kH, kW, T_out, ups, stride, pad_h, pad_w, H_out, W_out;
};
bool tcx_conv_mfma_supported(const TcxConvArgs& a);
int tcx_conv_mfma_launch(const TcxConvArgs& a, hipStream_t st);

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


// Sum over aligned groups of 8 (or 16) lanes with DPP moves (plain VALU) instead of __shfl_xor (ds_bpermute: an LDS round
// trip and a wait per step).  After the two quad steps every lane of a quad holds the quad's sum, so the half-row /
// row mirrors only have to reach ANY lane of the other quad / half: the result is bit-identical to the xor-1,2,4(,8) tree.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror: lane i <-> 7 - i of each 8-lane half row
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v = group8_sum(v);
    v += dpp_move<0x140>(v);     // row_mirror: lane i <-> 15 - i of each 16-lane row
    return v;
}

// Sum over the 64 lanes, result in every lane: DPP inside rows of 16, then v_permlane16_swap / v_permlane32_swap (gfx950)
// between rows — no LDS round trips (the __shfl_xor form is six ds_bpermute + waits).
__device__ __forceinline__ float wave_sum(float v) {
    v = group16_sum(v);
    {   // rows 0<->1 and 2<->3: after the swap the pair (a, b) holds this row's and the partner row's sums
        const uint32_t u = __float_as_uint(v);
        auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    {
        const uint32_t u = __float_as_uint(v);
        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    return v;
}

// Bijective XCD-aware block remap: blocks that share `orig % 8` run on one XCD (observed
// round-robin dispatch); give each XCD a contiguous range of logical ids.  Speed only.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t orig, uint32_t nwg) {
    const uint32_t xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}
