// A plain C++ host program against libtcx_hip.so — no Python, no torch: the C ABI of include/tcx_hip.h used the way a foreign host
// (the reference's own process, a C++ server) would.  Device buffers come from hipMalloc; one GEMM with the bias + GELU epilogue and one
// LayerNorm + modulate are checked against straightforward host loops.  Built and run by tests/test_cabi.py::test_c_host_program (GPU).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tcx_hip.h"

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float frand(uint32_t& s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main() {
    if (tcx_version() < 1) { printf("bad version\n"); return 1; }
    int32_t info[4];
    if (tcx_device_info(0, info) != TCX_OK) { printf("device_info: %s\n", tcx_last_error_string()); return 1; }
    printf("device: %d CUs, %d B LDS, wave %d, gfx%d\n", info[0], info[1], info[2], info[3]);
    hipStream_t st;
    HIPCHK(hipStreamCreate(&st));
    uint32_t seed = 7;
    // ---- GEMM: y = gelu_tanh(x w^T + b), M = 300 (ragged tile), N = 264, K = 200 (K-tail instantiation)
    const int M = 300, N = 264, K = 200;
    std::vector<uint16_t> x(M * K), w(N * K), b(N), y(M * N);
    for (auto& v : x) v = f2bf(frand(seed));
    for (auto& v : w) v = f2bf(frand(seed) / std::sqrt((float)K));
    for (auto& v : b) v = f2bf(frand(seed));
    void *dx, *dw, *db, *dy;
    HIPCHK(hipMalloc(&dx, x.size() * 2)); HIPCHK(hipMalloc(&dw, w.size() * 2)); HIPCHK(hipMalloc(&db, b.size() * 2)); HIPCHK(hipMalloc(&dy, y.size() * 2));
    HIPCHK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
    int rc = tcx_gemm_bf16(dx, dw, db, dy, M, N, K, K, N, 0, TCX_GEMM_BIAS_GELU, nullptr, 0, 0, nullptr, nullptr, 0, 0, 0, st);
    if (rc != TCX_OK) { printf("gemm rc %d: %s\n", rc, tcx_last_error_string()); return 1; }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = bf2f(b[n]);
            for (int k = 0; k < K; ++k) acc += (double)bf2f(x[m * K + k]) * bf2f(w[n * K + k]);
            const double u = 0.7978845608028654 * (acc + 0.044715 * acc * acc * acc);
            const double ref = 0.5 * acc * (1.0 + std::tanh(u));
            const double err = std::fabs(bf2f(y[m * N + n]) - ref), tol = std::fabs(ref) / 128.0 + 2e-3;
            if (err > tol) { printf("gemm mismatch at (%d,%d): %g vs %g\n", m, n, bf2f(y[m * N + n]), ref); return 1; }
            if (err > worst) worst = err;
        }
    printf("gemm %dx%dx%d + bias + gelu: worst |err| %.3g\n", M, N, K, worst);
    // ---- a rejected call reports through the error string and the return code
    rc = tcx_gemm_bf16(dx, dw, db, dy, M, N + 1, K, K, N, 0, TCX_GEMM_BIAS, nullptr, 0, 0, nullptr, nullptr, 0, 0, 0, st);
    if (rc != TCX_E_SHAPE && rc != TCX_E_ALIGN) { printf("expected a shape / alignment error for N %% 8 != 0, got %d\n", rc); return 1; }
    printf("rejected call: rc %d, \"%s\"\n", rc, tcx_last_error_string());
    // ---- LayerNorm + modulate: y = LN(x) gamma + beta, then * (1 + scale) + shift, C = 256, 37 rows, B = 2
    const int B = 2, R = 37, C = 256;
    std::vector<uint16_t> lx(B * R * C), g(C), be(C), sh(B * C), sc(B * C), ly(B * R * C);
    for (auto& v : lx) v = f2bf(frand(seed) * 2.0f + 0.3f);
    for (auto& v : g) v = f2bf(1.0f + 0.2f * frand(seed));
    for (auto& v : be) v = f2bf(0.1f * frand(seed));
    for (auto& v : sh) v = f2bf(0.5f * frand(seed));
    for (auto& v : sc) v = f2bf(0.5f * frand(seed));
    void *dlx, *dg, *dbe, *dsh, *dsc, *dly;
    HIPCHK(hipMalloc(&dlx, lx.size() * 2)); HIPCHK(hipMalloc(&dg, g.size() * 2)); HIPCHK(hipMalloc(&dbe, be.size() * 2));
    HIPCHK(hipMalloc(&dsh, sh.size() * 2)); HIPCHK(hipMalloc(&dsc, sc.size() * 2)); HIPCHK(hipMalloc(&dly, ly.size() * 2));
    HIPCHK(hipMemcpy(dlx, lx.data(), lx.size() * 2, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dg, g.data(), g.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dbe, be.data(), be.size() * 2, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dsh, sh.data(), sh.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsc, sc.data(), sc.size() * 2, hipMemcpyHostToDevice));
    rc = tcx_layernorm_modulate(dlx, dly, B, R, C, (int64_t)R * C, (int64_t)R * C, dg, dbe, dsh, dsc, nullptr, nullptr, C, 0, 1e-5f, st);
    if (rc != TCX_OK) { printf("layernorm rc %d: %s\n", rc, tcx_last_error_string()); return 1; }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(ly.data(), dly, ly.size() * 2, hipMemcpyDeviceToHost));
    worst = 0;
    for (int bi = 0; bi < B; ++bi)
        for (int r = 0; r < R; ++r) {
            const uint16_t* row = &lx[(bi * R + r) * C];
            double mean = 0, var = 0;
            for (int c = 0; c < C; ++c) mean += bf2f(row[c]);
            mean /= C;
            for (int c = 0; c < C; ++c) var += (bf2f(row[c]) - mean) * (bf2f(row[c]) - mean);
            const double rstd = 1.0 / std::sqrt(var / C + 1e-5);
            for (int c = 0; c < C; ++c) {
                const double ln = (bf2f(row[c]) - mean) * rstd * bf2f(g[c]) + bf2f(be[c]);
                const double ref = ln * (1.0 + bf2f(sc[bi * C + c])) + bf2f(sh[bi * C + c]);
                const double err = std::fabs(bf2f(ly[(bi * R + r) * C + c]) - ref), tol = std::fabs(ref) / 128.0 + 2e-3;
                if (err > tol) { printf("layernorm mismatch b %d r %d c %d: %g vs %g\n", bi, r, c, bf2f(ly[(bi * R + r) * C + c]), ref); return 1; }
                if (err > worst) worst = err;
            }
        }
    printf("layernorm + modulate [%d,%d,%d]: worst |err| %.3g\n", B, R, C, worst);
    printf("cabi_smoke ok\n");
    return 0;
}
