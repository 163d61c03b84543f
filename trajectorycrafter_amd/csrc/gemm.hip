// bf16 GEMM with fused epilogues for the transformer's Linear layers:
//     Y[M,N] = epilogue( X[M,K] . W[N,K]^T + bias[N] )           fp32 accumulate, bf16 in / out
// epilogues: bias | bias + GELU(tanh) | res + gate * (acc + bias)   (the gated residual of CogVideoXBlock,
// reference models/crosstransformer3d.py:245-248, 261-264).
//
// Structure (MI355X / gfx950 only): 256 x 256 output tile per 512-thread workgroup, K step 64, 8 waves as 2 (M) x 4 (N),
// each wave a 128 x 64 sub-tile = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 (128 VGPRs).  One workgroup per CU
// (128 KiB LDS = two K-tile buffers x four 16 KiB "half-tiles").
//
//  * Operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4), never through registers.  The LDS image is lane-
//    linear per wave instruction (1 KiB = 8 rows x 128 B), so the bank swizzle is applied to the SOURCE address:
//    16-byte slot s of row r holds k-slot s ^ ((r >> 1) & 7); ds_read_b128 fragment reads are then conflict-free
//    for the instruction's 16-lane groups (MI355X_MICROARCH.md, LDS table).
//  * A half-tile is what one QUADRANT of every wave needs: X half qm = rows {wr*128 + qm*64 + [0,64)} of both wave
//    rows, W half qn = columns {wc*64 + qn*32 + [0,32)} of all four wave columns.  A K-tile is computed in four phases,
//    one 64 x 32 quadrant each, order (0,0) (0,1) (1,1) (1,0): X half 0 is read in phase 0 only, W half 1 in phase 1,
//    X half 1 in phase 2, W half 0 in phases 0 and 3 — so each half-tile can be re-staged one phase after its last read
//    and exactly one half-tile is staged per phase.
//  * The two wave rows run one barrier apart (wr = 1 takes one extra s_barrier before the loop, wr = 0 one after it):
//    while one wave of a SIMD issues its 16 MFMAs the other reads fragments and issues the LDS-DMA.
//  * Ordering, per phase and wave:  ds_reads ; stage (2 glds) ; [vmcnt(6) in phases 3 and 7] ; lgkmcnt(0) ; s_barrier ;
//    16 MFMA ; s_barrier.  (Measured with tools/exp_gemm.sh: without the stagger -14 %; lgkmcnt(0) moved behind the
//    barrier — which would break the WAR argument below — gains nothing; s_setprio around the MFMAs -0.5..-1 %.)  With L_p / M_p the load / MFMA segments and k_j the j-th barrier, wave row 0 runs L_p in
//    (k_2p, k_2p+1) and wave row 1 in (k_2p+1, k_2p+2):
//      WAR: reads of phase p have returned (lgkmcnt(0)) before k_2p+2 at the latest; the earliest stage of phase q is
//           issued after k_2q  ->  re-staging in phase q >= p + 1 is safe.
//      RAW: a wave's LDS-DMA is complete when ITS vmcnt says so; all waves pass their phase-w wait before k_2w+2, the
//           earliest read of phase r is after k_2r  ->  data waited for in phase w is readable from phase w + 1.
//    Stage order (E / O = even / odd K-tile buffer): ph0 W0->O(kt+1) | ph1 X0->E(kt+2) | ph2 W1->E | ph3 X1->E |
//    ph4 W0->E | ph5 X0->O(kt+3) | ph6 W1->O | ph7 X1->O.  vmcnt(6) in phase 3 leaves the three newest half-tiles
//    (ph1..3) in flight and retires everything up to ph0's W0->O: the odd tile is complete for phases 4-7; likewise
//    phase 7 completes the even tile for the next iteration.  Never vmcnt(0) inside the loop.
//  * XCD-aware, bijective block remap; within an XCD's chunk tiles are ordered 4 (M) x all (N) so the 32 workgroups
//    sharing an L2 reuse each X tile 8x and each W tile 4x.
//  * Rows >= M / columns >= N are loaded from the last valid row of X / W and masked at the store; N % 8 == 0 and
//    K % 8 == 0 are required.  K % 128 == 0 (every Linear of the 5B transformer) runs the plain loop; any other K runs the
//    KTAIL instantiation: the K range is rounded up to an even number of 64-deep K-tiles and every 16-byte k-slot at or
//    beyond K is fetched from a 16-byte page of zeros in the code object instead (LDS-DMA takes a per-lane source
//    address, so "zero-fill" is just another address; two compares and selects per staged piece).
#include "tcx_common.h"
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * BK * 2;   // 16 KiB: 128 rows x 64 bf16
constexpr int BUF = 4 * HALF;        // X0 X1 W0 W1
constexpr int LDS_BYTES = 2 * BUF;
#ifndef TCX_GEMM_GM
#define TCX_GEMM_GM 4
#endif
constexpr int GM = TCX_GEMM_GM;      // M-tiles per band of the tile order

struct GemmParams {
    const uint16_t *x, *w, *bias;
    uint16_t* y;
    const uint16_t *res, *gate_v, *gate_t;
    int64_t M, ldx, ldy, ldres, gate_stride_b, y_stride_b, res_stride_b;
    int32_t N, K, rows_per_batch, text_len, mt, nt;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// gelu_tanh(x) = 0.5 x (1 + tanh(u)) = x / (1 + 2^(-2 u log2 e)),  u = sqrt(2/pi) (x + 0.044715 x^3): one exp2 and one
// reciprocal (1 ulp each; the result is rounded to bf16, 2^-9 relative) instead of an exp, an IEEE division and a dozen
// VALU ops per element of the 128-per-lane epilogue.  2^+inf -> x / inf = -0 for x -> -inf, 2^-inf -> x.
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float e = __builtin_amdgcn_exp2f(c * __builtin_fmaf(0.044715f * x * x, x, x));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

#define TCX_SB() __builtin_amdgcn_sched_barrier(0)

// what a k-slot at or beyond K reads (KTAIL): zeros contribute nothing to the accumulators
__device__ __attribute__((aligned(16))) const uint32_t g_zero_page[4] = {0u, 0u, 0u, 0u};

template <int EPI, bool KTAIL>
__global__ __launch_bounds__(512) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    // ---- tile of this workgroup ----
    const int t = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int band = t / (GM * p.nt), rr = t - band * (GM * p.nt);
    const int gme = min(GM, p.mt - band * GM);
    const int tm = band * GM + rr % gme, tn = rr / gme;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int KT = KTAIL ? ((p.K + 2 * BK - 1) / (2 * BK)) * 2 : p.K / BK;     // even number of K-tiles

    // ---- staging addresses: wave `wid` writes the 1-KiB pieces 2*wid, 2*wid+1 of every half-tile ----
    const uint16_t* px[2][2];
    const uint16_t* pw[2][2];
    int kofs[2];                                                        // KTAIL: first k of the lane's slot inside a K-tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = (wid * 2 + j) * 8 + (lane >> 3);                 // row of the half-tile
        const int ksl = (lane & 7) ^ ((lr >> 1) & 7);                   // source k-slot of LDS slot (lane & 7)
        kofs[j] = ksl * 8;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int64_t row = m0 + (lr >> 6) * 128 + q * 64 + (lr & 63);
            row = row < p.M ? row : p.M - 1;
            px[q][j] = p.x + row * p.ldx + ksl * 8;
            int col = n0 + (lr >> 5) * 64 + q * 32 + (lr & 31);
            col = col < p.N ? col : p.N - 1;
            pw[q][j] = p.w + (int64_t)col * p.K + ksl * 8;
        }
    }
    auto stage = [&](auto hsel, int buf, int kt) __attribute__((always_inline)) {
        constexpr int H = decltype(hsel)::value;                        // 0 X0, 1 X1, 2 W0, 3 W1
        kt = kt < KT ? kt : KT - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint16_t* src = (H < 2 ? px[H & 1][j] : pw[H & 1][j]) + (int64_t)kt * BK;
            if constexpr (KTAIL) {
                if (kt * BK + kofs[j] >= p.K) src = reinterpret_cast<const uint16_t*>(g_zero_page);
            }
            char* dst = lds + buf * BUF + H * HALF + (wid * 2 + j) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    using H2 = std::integral_constant<int, 2>;
    using H3 = std::integral_constant<int, 3>;

    // ---- fragment read addresses ----
    const int fi = lane & 15, fg = lane >> 4, fx = (fi >> 1) & 7;
    const int xo0 = (fg ^ fx) * 16, xo1 = xo0 ^ 64;
    const int ax = (wr * 64 + fi) * 128, aw = (wc * 32 + fi) * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[4][2], wf[2][2];

    auto read_x = [&](int buf, int qm) __attribute__((always_inline)) {
        const char* base = lds + buf * BUF + qm * HALF + ax;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            xf[tt][0] = *reinterpret_cast<const bf16x8*>(base + tt * 2048 + xo0);
            xf[tt][1] = *reinterpret_cast<const bf16x8*>(base + tt * 2048 + xo1);
        }
    };
    auto read_w = [&](int buf, int qn) __attribute__((always_inline)) {
        const char* base = lds + buf * BUF + (2 + qn) * HALF + aw;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            wf[u][0] = *reinterpret_cast<const bf16x8*>(base + u * 2048 + xo0);
            wf[u][1] = *reinterpret_cast<const bf16x8*>(base + u * 2048 + xo1);
        }
    };
    auto mfma_quadrant = [&](auto qmsel, auto qnsel) __attribute__((always_inline)) {
        constexpr int QM = decltype(qmsel)::value, QN = decltype(qnsel)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[QM * 4 + tt][QN * 2 + u] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][ks], xf[tt][ks], acc[QM * 4 + tt][QN * 2 + u], 0, 0, 0);
    };
    // one phase: P = 0..7 (P >> 2 = buffer being computed), kt2 = even K-tile of this iteration
    // LAST = the tile's last iteration: only phase 0 still has something to stage (W half 0 of the odd K-tile); nothing
    // is fetched past the end of K, and the single wait that remains is a vmcnt(0) with nothing else in flight.
    auto phase = [&](auto psel, auto lastsel, int kt2) __attribute__((always_inline)) {
        constexpr int P = decltype(psel)::value, B = P >> 2, Q = P & 3;
        constexpr bool LAST = decltype(lastsel)::value;
        if constexpr (Q == 0) { read_w(B, 0); TCX_SB(); read_x(B, 0); }
        if constexpr (Q == 1) read_w(B, 1);
        if constexpr (Q == 2) read_x(B, 1);
        if constexpr (Q == 3) read_w(B, 0);
        TCX_SB();
        if constexpr (P == 0) stage(H2{}, 1, kt2 + 1);
        if constexpr (!LAST) {
            if constexpr (P == 1) stage(H0{}, 0, kt2 + 2);
            if constexpr (P == 2) stage(H3{}, 0, kt2 + 2);
            if constexpr (P == 3) stage(H1{}, 0, kt2 + 2);
            if constexpr (P == 4) stage(H2{}, 0, kt2 + 2);
            if constexpr (P == 5) stage(H0{}, 1, kt2 + 3);
            if constexpr (P == 6) stage(H3{}, 1, kt2 + 3);
            if constexpr (P == 7) stage(H1{}, 1, kt2 + 3);
        }
        TCX_SB();
        if constexpr (Q == 3 && !LAST) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if constexpr (P == 3 && LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TCX_SB();
        if constexpr (Q == 0) mfma_quadrant(H0{}, H0{});
        if constexpr (Q == 1) mfma_quadrant(H0{}, H1{});
        if constexpr (Q == 2) mfma_quadrant(H1{}, H1{});
        if constexpr (Q == 3) mfma_quadrant(H1{}, H0{});
        TCX_SB();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TCX_SB();
    };

    // ---- prologue: K-tile 0 complete, three half-tiles of K-tile 1 in flight ----
    stage(H0{}, 0, 0);
    stage(H1{}, 0, 0);
    stage(H2{}, 0, 0);
    stage(H3{}, 0, 0);
    stage(H0{}, 1, 1);
    stage(H3{}, 1, 1);
    stage(H1{}, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();      // wave row 1 runs one barrier behind
    TCX_SB();

    auto iteration = [&](auto lastsel, int kt2) __attribute__((always_inline)) {
        phase(std::integral_constant<int, 0>{}, lastsel, kt2);
        phase(std::integral_constant<int, 1>{}, lastsel, kt2);
        phase(std::integral_constant<int, 2>{}, lastsel, kt2);
        phase(std::integral_constant<int, 3>{}, lastsel, kt2);
        phase(std::integral_constant<int, 4>{}, lastsel, kt2);
        phase(std::integral_constant<int, 5>{}, lastsel, kt2);
        phase(std::integral_constant<int, 6>{}, lastsel, kt2);
        phase(std::integral_constant<int, 7>{}, lastsel, kt2);
    };
    // every iteration stages; past the end of K the last one re-loads the last K-tile into buffers nobody reads again
    // (keeps the vmcnt counts, 7 half-tiles of extra L2 traffic per tile) and the loads are drained here
    for (int kt2 = 0; kt2 < KT; kt2 += 2) iteration(std::false_type{}, kt2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing LDS-DMA must land before the LDS is released
    if (wr == 0) __builtin_amdgcn_s_barrier();      // balance the barrier count
    // A wave's vmcnt(0) covers only its OWN trailing LDS-DMA, whose pieces land in other waves' epilogue tiles (lds + wid * 16 KiB
    // spans both K-tile buffers): every wave must have drained before any wave writes its tile.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- epilogue: lane holds C[m = .. + fi][n = .. + 4 fg + 0..3] of each 16x16 tile ----
    // Branch-free loads (out-of-range rows / columns read a clamped, valid address; only the stores are predicated) so
    // that a row's residual and gate vectors are all in flight together instead of one L2 round trip each.
    int ncol[4];
    bool nok[4];
    float bv[4][4];
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
        const int n = n0 + wc * 64 + bq * 16 + fg * 4;
        nok[bq] = n < p.N;                                   // N % 4 == 0: a lane's four columns are in or out together
        ncol[bq] = nok[bq] ? n : p.N - 4;
        u32x2 bb = {0u, 0u};
        if (p.bias) bb = *reinterpret_cast<const u32x2*>(p.bias + ncol[bq]);
        bv[bq][0] = bf16lo(bb[0]); bv[bq][1] = bf16hi(bb[0]); bv[bq][2] = bf16lo(bb[1]); bv[bq][3] = bf16hi(bb[1]);
    }
    auto store_rows = [&](auto gated) __attribute__((always_inline)) {
        constexpr bool GATED = decltype(gated)::value;
        // row geometry of the lane's 8 rows; rows_per_batch > 0: row m = (batch b, row rb), y / res / gates are addressed
        // per batch (strided row ranges)
        bool mok[8];
        uint32_t bb[8], rbb[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const uint32_t mr = (uint32_t)m0 + wr * 128 + a * 16 + fi;          // M < 2^31 (checked by the host)
            mok[a] = mr < (uint32_t)p.M;
            const uint32_t m = mok[a] ? mr : (uint32_t)p.M - 1;
            bb[a] = 0;
            rbb[a] = m;
            if (p.rows_per_batch > 0) { bb[a] = m / (uint32_t)p.rows_per_batch; rbb[a] = m - bb[a] * (uint32_t)p.rows_per_batch; }
        }
        // residual: all of the lane's vectors in flight at once (the fragment registers of the main loop are free now).
        // Default: fetched ROW-WISE (16 instructions of 8 rows x 128 B, 16 bytes per lane) into the wave's LDS tile, the
        // same image the result is written back through, and picked up from there in the accumulator layout.
        if constexpr (EPI == 2) {
            {
                const int rl = lane >> 3, ch = lane & 7;
                int nc8 = n0 + wc * 64 + ch * 8;
                nc8 = nc8 < p.N ? nc8 : p.N - 8;
                u32x4 rr[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    uint32_t m = (uint32_t)m0 + wr * 128 + i * 8 + rl;
                    m = m < (uint32_t)p.M ? m : (uint32_t)p.M - 1;
                    uint32_t b = 0, rb = m;
                    if (p.rows_per_batch > 0) { b = m / (uint32_t)p.rows_per_batch; rb = m - b * (uint32_t)p.rows_per_batch; }
                    rr[i] = *reinterpret_cast<const u32x4*>(p.res + (int64_t)b * p.res_stride_b + (int64_t)rb * p.ldres + nc8);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = i * 8 + rl;
                    *reinterpret_cast<u32x4*>(lds + wid * 16384 + row * 128 + ((ch ^ (row & 7)) << 4)) = rr[i];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile now holds the residual; each element is read
            }                                                        // and then overwritten with its result by the same lane
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            u32x2 gv[4];
            if constexpr (EPI == 2 && GATED) {
                const uint16_t* gate = (rbb[a] < (uint32_t)p.text_len ? p.gate_t : p.gate_v) + (int64_t)bb[a] * p.gate_stride_b;
#pragma unroll
                for (int bq = 0; bq < 4; ++bq) gv[bq] = *reinterpret_cast<const u32x2*>(gate + ncol[bq]);
            }
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[a][bq][j] + bv[bq][j];
                if constexpr (EPI == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_tanh_f(v[j]);
                }
                if constexpr (EPI == 2) {
                    const u32x2 rvv = *reinterpret_cast<const u32x2*>(lds + wid * 16384 + (a * 16 + fi) * 128 +
                                                                      (((bq * 2 + (fg >> 1)) ^ ((a * 16 + fi) & 7)) << 4) + ((fg & 1) << 3));
                    const float r[4] = {bf16lo(rvv[0]), bf16hi(rvv[0]), bf16lo(rvv[1]), bf16hi(rvv[1])};
                    if constexpr (GATED) {
                        const float gg[4] = {bf16lo(gv[bq][0]), bf16hi(gv[bq][0]), bf16lo(gv[bq][1]), bf16hi(gv[bq][1])};
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = r[j] + gg[j] * v[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = r[j] + v[j];
                    }
                }
                u32x2 o;
                o[0] = pack_bf16(v[0], v[1]);
                o[1] = pack_bf16(v[2], v[3]);
                // through the wave's own 16 KiB of LDS (128 rows x 128 B; the main loop's buffers are free: every wave is past
                // its last fragment read after the final barrier): 16-byte chunk c of row r sits at chunk c ^ (r & 7)
                const int row = a * 16 + fi, chunk = bq * 2 + (fg >> 1);
                *reinterpret_cast<u32x2*>(lds + wid * 16384 + row * 128 + ((chunk ^ (row & 7)) << 4) + ((fg & 1) << 3)) = o;
            }
        }
        (void)mok;
        // read back row-wise: one instruction = 8 rows x 128 B, full cache lines, 16-byte stores (the direct form was 32 stores
        // of 8 bytes per lane, each instruction touching 16 rows x 32 B: store-issue bound)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int rl = lane >> 3, ch = lane & 7;
        const int ncol8 = n0 + wc * 64 + ch * 8;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 8 + rl;
            const u32x4 val = *reinterpret_cast<const u32x4*>(lds + wid * 16384 + row * 128 + ((ch ^ (row & 7)) << 4));
            const uint32_t m = (uint32_t)m0 + wr * 128 + row;
            if (m < (uint32_t)p.M && ncol8 < p.N) {
                uint32_t b = 0, rb = m;
                if (p.rows_per_batch > 0) { b = m / (uint32_t)p.rows_per_batch; rb = m - b * (uint32_t)p.rows_per_batch; }
                *reinterpret_cast<u32x4*>(p.y + (int64_t)b * p.y_stride_b + (int64_t)rb * p.ldy + ncol8) = val;
            }
        }
    };
    if (EPI == 2 && p.gate_v) store_rows(std::true_type{});
    else store_rows(std::false_type{});
}

// ---- skinny GEMM: M <= 8 rows (the AdaLN modulation and time-embedding Linears: M = batch = 2, 85 launches per step) -----------
// A 256 x 256 MFMA tile is 99 % padding there and costs ~15 us of pure latency (prologue + 8 K-tiles + epilogue).  This is a
// weight-streaming kernel instead: 8 lanes share an output column (lane & 7 owns every 8th 16-byte chunk of the weight row:
// a wave-instruction reads 8 rows x 128 contiguous bytes), x sits in LDS, products on v_dot2c_f32_bf16 (fp32 accumulate),
// the 8 partial sums of a column are folded with DPP adds.  HBM-bound: N K 2 bytes of weights once.
constexpr int kSkinnyMaxM = 8;
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char xl[];               // x rows: [M][K] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kp = lane & 7, nc = lane >> 3;
    const int M = (int)p.M, kch = p.K >> 3;                                  // 16-byte chunks per row
    for (int i = tid; i < M * kch; i += 256) {
        const int m = i / kch, c = i - m * kch;
        reinterpret_cast<u32x4*>(xl)[i] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)m * p.ldx + 8 * c);
    }
    __syncthreads();
    const int n = ((int)blockIdx.x * 4 + wave) * 8 + nc;
    const int nn = n < p.N ? n : p.N - 1;
    const uint16_t* wr = p.w + (int64_t)nn * p.K;
    float acc[kSkinnyMaxM];
#pragma unroll
    for (int m = 0; m < kSkinnyMaxM; ++m) acc[m] = 0.f;
    for (int c0 = kp; c0 < kch; c0 += 32) {                                  // four weight chunks in flight per lane
        u32x4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 8 * u;
            wv[u] = c < kch ? *reinterpret_cast<const u32x4*>(wr + 8 * c) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 8 * u;
            if (c < kch) {
#pragma unroll
                for (int m = 0; m < kSkinnyMaxM; ++m) {
                    if (m < M) {
                        const u32x4 xv = *reinterpret_cast<const u32x4*>(xl + ((size_t)m * kch + c) * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t xa = xv[j], wa = wv[u][j];
                            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[m]) : "v"(xa), "v"(wa));
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < kSkinnyMaxM; ++m) acc[m] = group8_sum(acc[m]);
    if (kp == 0 && n < p.N) {
        const float b = p.bias ? bf16_bits_to_f32(p.bias[n]) : 0.f;
#pragma unroll
        for (int m = 0; m < kSkinnyMaxM; ++m)
            if (m < M) p.y[(int64_t)m * p.ldy + n] = (uint16_t)(pack_bf16(acc[m] + b, 0.f) & 0xffff);
    }
}

template <int EPI, bool KTAIL>
int launch_gemm_k(const GemmParams& p, hipStream_t st) {
    static TcxPerDeviceOnce lds_attr;          // hipFuncSetAttribute is per device: once per (kernel, device), thread-safe
    const int rc = tcx_ensure_dynamic_lds(lds_attr, reinterpret_cast<const void*>(&gemm_kernel<EPI, KTAIL>), LDS_BYTES, "tcx_gemm_bf16");
    if (rc != TCX_OK) return rc;
    hipLaunchKernelGGL((gemm_kernel<EPI, KTAIL>), dim3((unsigned)(p.mt * p.nt)), dim3(512), LDS_BYTES, st, p);
    TCX_LAUNCH_RET();
}

template <int EPI>
int launch_gemm(const GemmParams& p, hipStream_t st) {
    return p.K % (2 * BK) == 0 ? launch_gemm_k<EPI, false>(p, st) : launch_gemm_k<EPI, true>(p, st);
}

}  // namespace

extern "C" int tcx_gemm_bf16(const void* x, const void* w, const void* bias, void* y, int64_t M, int32_t N, int32_t K,
                             int64_t ldx, int64_t ldy, int64_t y_stride_b, int32_t epilogue, const void* res, int64_t ldres,
                             int64_t res_stride_b, const void* gate_v, const void* gate_t, int64_t gate_stride_b,
                             int32_t rows_per_batch, int32_t text_len, void* stream) {
    TCX_CHECK(x && w && y, TCX_E_NULL, "tcx_gemm_bf16: null x / w / y");
    TCX_CHECK(M < (1ll << 31), TCX_E_SHAPE, "tcx_gemm_bf16: M=%lld exceeds 2^31 rows", (long long)M);
    TCX_CHECK(M > 0 && N > 0 && K > 0, TCX_E_SHAPE, "tcx_gemm_bf16: empty shape M=%lld N=%d K=%d", (long long)M, N, K);
    TCX_CHECK(N % 8 == 0 && K % 8 == 0, TCX_E_SHAPE, "tcx_gemm_bf16: needs N %% 8 == 0 and K %% 8 == 0 (N=%d K=%d)", N, K);
    TCX_CHECK(ldx >= K && ldy >= N && ldx % 8 == 0 && ldy % 8 == 0, TCX_E_SHAPE, "tcx_gemm_bf16: bad leading dimensions ldx=%lld ldy=%lld", (long long)ldx, (long long)ldy);
    TCX_CHECK(tcx_aligned16(x) && tcx_aligned16(w) && tcx_aligned16(y) && tcx_aligned16(bias), TCX_E_ALIGN, "tcx_gemm_bf16: pointers must be 16-byte aligned");
    TCX_CHECK(epilogue >= 0 && epilogue <= 2, TCX_E_SHAPE, "tcx_gemm_bf16: unknown epilogue %d", epilogue);
    const int64_t mt = (M + BM - 1) / BM;
    const int32_t ntile = (N + BN - 1) / BN;
    TCX_CHECK(mt * ntile < (1ll << 31), TCX_E_SHAPE, "tcx_gemm_bf16: too many tiles");
    GemmParams p{};
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w; p.bias = (const uint16_t*)bias; p.y = (uint16_t*)y;
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldy = ldy; p.mt = (int32_t)mt; p.nt = ntile;
    TCX_CHECK(rows_per_batch >= 0 && (rows_per_batch == 0 || M % rows_per_batch == 0), TCX_E_SHAPE,
              "tcx_gemm_bf16: M=%lld is not a multiple of rows_per_batch=%d", (long long)M, rows_per_batch);
    TCX_CHECK(rows_per_batch > 0 || y_stride_b == 0, TCX_E_SHAPE, "tcx_gemm_bf16: y_stride_b needs rows_per_batch");
    TCX_CHECK(y_stride_b % 8 == 0 && res_stride_b % 8 == 0, TCX_E_ALIGN, "tcx_gemm_bf16: batch strides must be multiples of 4 elements");
    p.rows_per_batch = rows_per_batch; p.y_stride_b = rows_per_batch > 0 ? y_stride_b : 0;
    if (epilogue == TCX_GEMM_GATED_RESIDUAL) {
        TCX_CHECK(res, TCX_E_NULL, "tcx_gemm_bf16: the gated-residual epilogue needs res");
        TCX_CHECK(ldres >= N && ldres % 8 == 0 && tcx_aligned16(res), TCX_E_SHAPE, "tcx_gemm_bf16: bad res layout");
        TCX_CHECK((gate_v == nullptr) == (gate_t == nullptr), TCX_E_NULL, "tcx_gemm_bf16: give both gates or none");
        if (gate_v) {
            TCX_CHECK(rows_per_batch > 0 && text_len >= 0 && text_len <= rows_per_batch && gate_stride_b % 4 == 0,
                      TCX_E_SHAPE, "tcx_gemm_bf16: bad gate geometry rows_per_batch=%d text_len=%d", rows_per_batch, text_len);
            TCX_CHECK(tcx_aligned16(gate_v) && tcx_aligned16(gate_t), TCX_E_ALIGN, "tcx_gemm_bf16: gates must be 16-byte aligned");
        }
        p.res = (const uint16_t*)res; p.ldres = ldres; p.gate_v = (const uint16_t*)gate_v; p.gate_t = (const uint16_t*)gate_t;
        p.gate_stride_b = gate_stride_b; p.text_len = text_len; p.res_stride_b = rows_per_batch > 0 ? res_stride_b : 0;
    }
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == TCX_GEMM_BIAS && M <= kSkinnyMaxM && K % 8 == 0 && rows_per_batch == 0 && (size_t)M * K * 2 <= 64 * 1024) {
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3((unsigned)((N + 31) / 32)), dim3(256), (size_t)M * K * 2, st, p);
        TCX_LAUNCH_RET();
    }
    switch (epilogue) {
        case TCX_GEMM_BIAS: return launch_gemm<0>(p, st);
        case TCX_GEMM_BIAS_GELU: return launch_gemm<1>(p, st);
        default: return launch_gemm<2>(p, st);
    }
}
