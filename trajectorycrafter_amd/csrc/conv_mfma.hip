// bf16 implicit-GEMM convolution on the GEMM pipeline of gemm.hip: the VAE's 3x3x3 causal convs, 3x3 upsample / downsample convs and
// 1x1x1 shortcuts with Cin % 128 == 0 (the decoder's and encoder's 128 / 256 / 512-channel layers = 99 % of the VAE's FLOP).
// Replaces reference models/autoencoder_magvit.py:76-163 (CogVideoXCausalConv3d), :620-630 (upsample + Conv2d).
//
//     Y[m, co] = bias[co] (+ res[m, co]) + sum_k Xg[m, k] * W[co, k]        m = (n, t, oy, ox),  k = (tap, ci),  tap = (dt, dy, dx)
//
// Same machinery as gemm.hip (read its header for the 8-phase schedule and the WAR / RAW argument; they carry over unchanged):
// 512 threads = 8 waves, v_mfma_f32_16x16x32_bf16, wave sub-tile 128 x 64 = 8 x 4 accumulators, K step 64, two K-tile buffers
// in LDS, operands HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) with the bank swizzle on the SOURCE address, one quadrant
// (64 x 32) of every wave per phase, the two wave halves (waves 0-3 / 4-7 = the two waves of each SIMD) one barrier apart,
// counted vmcnt (never 0 in the loop), results / residual through the wave's 16 KiB LDS tile in 16-byte row-wise accesses.
//
// What is different:
//  * The X operand is GATHERED: row m of K-tile kt is 128 contiguous bytes of the input pixel that tap(kt) of output
//    position m reads (Cin % 64 == 0: a K-tile never straddles two taps), or 128 bytes of a ZERO PAGE in the code object when
//    that pixel is spatial padding.  LDS-DMA takes a per-lane source address, so padding, the causal temporal context (cache
//    frames or the replicated first frame), the nearest x2 upsample (>> ups), the stride-2 downsample and the temporal frame
//    map are all just address arithmetic; nothing padded / upsampled / im2col'ed exists in memory.
//  * Address generation is per ROW, not per lane-piece: a wave stages 2 x XP pieces of 8 rows per K-tile = at most 64 distinct
//    rows, so lane L owns row L of them, recomputes that row's pixel pointer only when the tap changes (every Cin / 64
//    K-tiles, ~25 VALU), and the staging lanes then fetch the pointers of the rows they stage (row (lane >> 3) of each piece)
//    with two ds_bpermute_b32 per piece, keeping them in registers until the next tap.
//  * Two tile shapes: Cout >= 256 -> 256 (m) x 256 (co) exactly like gemm.hip (waves 2 x 4); Cout == 128 (the 480x720 stage,
//    44 % of the decoder's FLOP) -> 512 (m) x 128 (co) (waves 4 x 2): the wave sub-tile, hence the LDS-read : MFMA ratio, stays
//    128 x 64 (a 256 x 128 tile of 64 x 64 wave tiles would need 1.4x the LDS reads per MFMA and sit at the LDS roof).  Its
//    X half-tile is 32 KiB (4 pieces per wave) and its W half-tile 8 KiB (1 piece): the per-phase staging is 2, 3, 2, 3 pieces
//    instead of 2, 2, 2, 2 and the in-loop wait is vmcnt(7); LDS = 2 x (64 + 16) KiB = all 160 KiB of the CU.
#include "tcx_common.h"
#include <type_traits>

int tcx_conv_mfma_launch(const TcxConvArgs& a, hipStream_t st);

namespace {

constexpr int BK = 64;
// what a padding pixel reads: Cin <= 1024 channels of zeros (offset ci0 + k-slot inside the page)
__device__ __attribute__((aligned(16))) const uint32_t g_conv_zero_page[512] = {};

struct ConvGemmParams {
    const uint16_t *x, *cache, *w, *bias, *res;
    uint16_t* y;
    const int32_t* t_map;
    int64_t M, frame_elems;
    int32_t T_in, W_in, Cin, Cout, kT, kH, kW, T_out, H, W, ups, LH, LW, stride, pad_h, pad_w, Ktot, frames, mt, nt;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

#define TCX_SB() __builtin_amdgcn_sched_barrier(0)
template <int I>
using IC = std::integral_constant<int, I>;

template <int WM, int WN, bool RES>
__global__ __launch_bounds__(512) void conv_mfma_kernel(const ConvGemmParams p) {
    static_assert(WM * WN == 8 && (WM == 2 || WM == 4), "8 waves as 2 x 4 or 4 x 2");
    constexpr int BM = WM * 128, BN = WN * 64;
    constexpr int XP = WM, WP = WN / 2;                  // LDS-DMA pieces (1 KiB = 8 rows x 128 B) per wave per X / W half-tile
    constexpr int XA = XP == 4 ? 2 : XP;                 // X pieces staged in the half's first slot (the rest + W in the next)
    constexpr int XH = WM * 64 * 128, WH = WN * 32 * 128, BUF = 2 * XH + 2 * WH;
    constexpr int S13 = XA, S24 = (XP - XA) + WP;        // pieces per staging slot: X0a | X0b + W1 | X1a | X1b + W0
    constexpr int INFLIGHT = S13 + S24 + S13;            // the three newest slots stay in flight across the phase-3 / 7 wait
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WN, wc = wid % WN;

    // ---- tile of this workgroup ----
    // Output frames (n, t) are tiled separately (tpf tiles of BM positions per frame, the last one ragged) and the tile order is
    // co-tile fastest, then FRAME, then position: with the XCD remap (a contiguous range of the order per XCD) one XCD owns a
    // spatial band across all frames, and the CUs working side by side read the same input rows — the 3 x 3 x 3 taps of
    // neighbouring (t, y) tiles — out of that XCD's L2.  (Raster order gave every XCD a different frame: each input frame was then
    // fetched by the three XCDs whose output frames read it, 3.6x the algorithmic bytes at the memory side.)
    const int t = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int L = t / p.nt, tn = t - L * p.nt;
    const int tj = L / p.frames, tf = L - tj * p.frames; // position tile tj of output frame tf = n * T_out + t
    const int HW = p.H * p.W;
    const int l0 = tj * BM;                              // first position of the tile inside its frame
    const int64_t m0 = (int64_t)tf * HW + l0;
    const int n0 = tn * BN;
    const int KT = p.Ktot / BK;                          // even (host-checked)

    // ---- gather state: lane L owns row L of the <= 64 rows this wave stages per K-tile ----
    // row L = piece (L >> 3) of the wave's 2 * XP X pieces (half q = piece / XP, index j = piece % XP), row (L & 7) of it.
    // All offsets are 32-bit element offsets from the lane's batch item (host-checked: (T_in + kT) frames < 2^31 elements).
    int g_oyp, g_oxp, g_t;                               // oy * stride - pad_h, ox * stride - pad_w, source frame
    const uint16_t *g_xn, *g_cn;                         // x / cache of the row's batch item
    {
        const int piece = (lane >> 3) & (2 * XP - 1);
        const int gq = piece / XP, gj = piece % XP;
        const int lr = (wid * XP + gj) * 8 + (lane & 7);
        int loc = l0 + (lr >> 6) * 128 + gq * 64 + (lr & 63);
        loc = loc < HW ? loc : HW - 1;                   // rows past the frame gather a valid pixel; they are masked at the store
        const int oy = loc / p.W, ox = loc - oy * p.W;
        const int64_t n = tf / p.T_out;
        const int tt = tf - (int)n * p.T_out;
        g_t = (p.kT == 1 && p.t_map) ? p.t_map[tt] : tt; // t_map exists only for kT == 1 (upsample conv: nearest in time)
        g_oyp = oy * p.stride - p.pad_h;
        g_oxp = ox * p.stride - p.pad_w;
        g_xn = p.x + n * p.T_in * p.frame_elems;
        g_cn = p.cache ? p.cache + n * (p.kT - 1) * p.frame_elems : nullptr;
    }
    int tap_dt = 0, tap_dy = 0, tap_dx = 0;              // wave-uniform: the tap of the K-tile slot 1 last staged
    auto tap_pointer = [&]() __attribute__((always_inline)) -> uint64_t {
        const int iy = g_oyp + tap_dy, ix = g_oxp + tap_dx;
        const bool ok = (unsigned)iy < (unsigned)p.LH && (unsigned)ix < (unsigned)p.LW;
        const int li = g_t + tap_dt;                     // logical input frame: kT - 1 context frames first
        const bool ctx = li < p.kT - 1;                  // previous chunk's last frames (cache) or, first chunk, frame 0 replicated
        const uint16_t* base = (ctx && g_cn) ? g_cn : g_xn;
        const int frame = ctx ? (g_cn ? li : 0) : li - (p.kT - 1);
        const int off = frame * (int)p.frame_elems + ((iy >> p.ups) * p.W_in + (ix >> p.ups)) * p.Cin;
        return reinterpret_cast<uint64_t>(ok ? base + off : reinterpret_cast<const uint16_t*>(g_conv_zero_page));
    };
    uint64_t g_nxt = tap_pointer();                      // this lane's row at the current tap (pixel, or the zero page)
    int ci0 = 0;                                         // first channel of the K-tile being staged (slots 1..4 of a K-tile are
                                                         // consecutive in time, so one value serves all four)

    // ---- staging ----
    const uint16_t* pw[2][WP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        {
            const int lr = (wid * WP + j) * 8 + (lane >> 3);
            const int ksl = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int col = n0 + (lr >> 5) * 64 + q * 32 + (lr & 31);
                col = col < p.Cout ? col : p.Cout - 1;
                pw[q][j] = p.w + (int64_t)col * p.Ktot + ksl * 8;
            }
        }
    }
    int kofx[XP];                                        // the lane's k-slot inside a 64-deep K-tile (source-side swizzle)
#pragma unroll
    for (int j = 0; j < XP; ++j) {
        const int lr = (wid * XP + j) * 8 + (lane >> 3);
        kofx[j] = ((lane & 7) ^ ((lr >> 1) & 7)) * 8;
    }
    const int perm_base = (lane >> 3) * 4;               // ds_bpermute byte address of row (lane >> 3) of piece 0

    // Row pointers of the pieces this lane stages: row (lane >> 3) of X piece (q, j) is owned by lane (q * XP + j) * 8 + (lane >> 3);
    // fetched with two ds_bpermute_b32 per piece WHEN THE TAP CHANGES (every Cin / 64 K-tiles) and kept in registers in between
    // (fetching them for every K-tile cost 10 % on the 256 x 256 shape: the DMA's address then waits for an LDS round trip).
    const uint16_t* xptr[2][XP];
    auto refresh_rows = [&](auto qsel) __attribute__((always_inline)) {
        constexpr int Q = decltype(qsel)::value;
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int a = perm_base + (Q * XP + j) * 32;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)g_nxt);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)(g_nxt >> 32));
            xptr[Q][j] = reinterpret_cast<const uint16_t*>(((uint64_t)hi << 32) | lo);
        }
    };
    refresh_rows(IC<0>{});
    refresh_rows(IC<1>{});
    bool tap_changed = false;                            // wave-uniform: slot 1 moved to a new tap, slot 3 has yet to follow
    auto issue_x = [&](auto qsel, auto jsel, int buf) __attribute__((always_inline)) {
        constexpr int Q = decltype(qsel)::value, J = decltype(jsel)::value;
        char* dst = lds + buf * BUF + Q * XH + (wid * XP + J) * 1024;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(xptr[Q][J] + (ci0 + kofx[J])), (lds_ptr_t)dst, 16, 0, 0);
    };
    auto stage_w = [&](auto qsel, auto jsel, int buf, int kt) __attribute__((always_inline)) {
        constexpr int Q = decltype(qsel)::value, J = decltype(jsel)::value;
        const uint16_t* src = pw[Q][J] + (int64_t)kt * BK;
        char* dst = lds + buf * BUF + 2 * XH + Q * WH + (wid * WP + J) * 1024;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    };
    // One staging slot of K-tile kt into buffer buf.  S = 1: X0 first pieces (+ the tap bookkeeping) | 2: X0 rest + W1 |
    // 3: X1 first pieces | 4: X1 rest + W0.  The four slots of a K-tile are consecutive in time (phases 1-4 / 5-8), so one
    // (tap, ci0) state serves them; X half 1 switches to a new tap's rows two phases after X half 0 (slot 3 vs slot 1).
    // Past the end of K the last K-tile is re-fetched into buffers nobody reads again (keeps the vmcnt counts).
    auto stage_slot = [&](auto ssel, int buf, int kt) __attribute__((always_inline)) {
        constexpr int S = decltype(ssel)::value;
        constexpr int Q = S <= 2 ? 0 : 1;
        if constexpr (S == 1) {                          // called with kt = 0, 1, 2, ...: advance (tap, ci0) by one K-tile
            if (kt > 0 && kt < KT) {
                ci0 += BK;
                if (ci0 >= p.Cin) {                      // wave-uniform branch, every Cin / 64 K-tiles: next tap
                    ci0 = 0;
                    if (++tap_dx == p.kW) { tap_dx = 0; if (++tap_dy == p.kH) { tap_dy = 0; ++tap_dt; } }
                    g_nxt = tap_pointer();
                    refresh_rows(IC<0>{});
                    tap_changed = true;
                }
            }
        }
        if constexpr (S == 3) {
            if (tap_changed) {
                refresh_rows(IC<1>{});
                tap_changed = false;
            }
        }
        kt = kt < KT ? kt : KT - 1;
        if constexpr (S == 1 || S == 3) {
            issue_x(IC<Q>{}, IC<0>{}, buf);
            if constexpr (XA > 1) issue_x(IC<Q>{}, IC<1>{}, buf);
        } else {
            if constexpr (XP > XA) {
                issue_x(IC<Q>{}, IC<XA>{}, buf);
                if constexpr (XP > XA + 1) issue_x(IC<Q>{}, IC<XA + 1>{}, buf);
            }
            stage_w(IC<1 - Q>{}, IC<0>{}, buf, kt);       // slot 2 stages W half 1, slot 4 W half 0
            if constexpr (WP > 1) stage_w(IC<1 - Q>{}, IC<1>{}, buf, kt);
        }
    };

    // ---- fragment reads / MFMA (gemm.hip) ----
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
        const char* base = lds + buf * BUF + qm * XH + ax;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            xf[tt][0] = *reinterpret_cast<const bf16x8*>(base + tt * 2048 + xo0);
            xf[tt][1] = *reinterpret_cast<const bf16x8*>(base + tt * 2048 + xo1);
        }
    };
    auto read_w = [&](int buf, int qn) __attribute__((always_inline)) {
        const char* base = lds + buf * BUF + 2 * XH + qn * WH + aw;
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
    // phase P = 0..7 (P >> 2 = buffer computed); stage order: ph0 slot 4 of K-tile kt2+1 -> odd buffer | ph1..4 slots 1..4 of
    // kt2+2 -> even buffer | ph5..7 slots 1..3 of kt2+3 -> odd buffer.  The phase-3 wait retires everything up to ph0's slot (the
    // odd tile is complete for phases 4-7), the phase-7 wait everything up to ph4's (the even tile for the next iteration).
    auto phase = [&](auto psel, int kt2) __attribute__((always_inline)) {
        constexpr int P = decltype(psel)::value, B = P >> 2, Q = P & 3;
        constexpr int S = P == 0 ? 4 : (P <= 4 ? P : P - 4);                    // staging slot of this phase
        const int skt = P == 0 ? kt2 + 1 : (P <= 4 ? kt2 + 2 : kt2 + 3);        // ... of which K-tile
        constexpr int sbuf = (P == 0 || P >= 5) ? 1 : 0;                        // ... into which buffer
        if constexpr (Q == 0) { read_w(B, 0); TCX_SB(); read_x(B, 0); }
        if constexpr (Q == 1) read_w(B, 1);
        if constexpr (Q == 2) read_x(B, 1);
        if constexpr (Q == 3) read_w(B, 0);
        TCX_SB();
        stage_slot(IC<S>{}, sbuf, skt);
        TCX_SB();
        if constexpr (Q == 3) {
            if constexpr (INFLIGHT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TCX_SB();
        if constexpr (Q == 0) mfma_quadrant(IC<0>{}, IC<0>{});
        if constexpr (Q == 1) mfma_quadrant(IC<0>{}, IC<1>{});
        if constexpr (Q == 2) mfma_quadrant(IC<1>{}, IC<1>{});
        if constexpr (Q == 3) mfma_quadrant(IC<1>{}, IC<0>{});
        TCX_SB();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TCX_SB();
    };

    // ---- prologue: K-tile 0 complete, slots 1..3 of K-tile 1 in flight ----
    stage_slot(IC<1>{}, 0, 0);
    stage_slot(IC<2>{}, 0, 0);
    stage_slot(IC<3>{}, 0, 0);
    stage_slot(IC<4>{}, 0, 0);
    stage_slot(IC<1>{}, 1, 1);
    stage_slot(IC<2>{}, 1, 1);
    stage_slot(IC<3>{}, 1, 1);
    if constexpr (INFLIGHT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wid >= 4) __builtin_amdgcn_s_barrier();          // waves 4-7 (the SIMD partners of 0-3) run one barrier behind
    TCX_SB();

    for (int kt2 = 0; kt2 < KT; kt2 += 2) {
        phase(IC<0>{}, kt2);
        phase(IC<1>{}, kt2);
        phase(IC<2>{}, kt2);
        phase(IC<3>{}, kt2);
        phase(IC<4>{}, kt2);
        phase(IC<5>{}, kt2);
        phase(IC<6>{}, kt2);
        phase(IC<7>{}, kt2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing LDS-DMA must land before the LDS is reused / released
    if (wid < 4) __builtin_amdgcn_s_barrier();           // balance the barrier count
    // vmcnt(0) drains only the wave's OWN trailing LDS-DMA, and those pieces land in the buffer regions that are OTHER waves'
    // epilogue tiles: every wave must have drained before any wave writes its tile (one workgroup-wide barrier, ~1 phase of wait
    // for waves 0-3).
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- epilogue (gemm.hip): lane holds C[m = .. + fi][co = .. + 4 fg + 0..3] of each 16 x 16 tile ----
    int ncol[4];
    float bv[4][4];
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
        const int n = n0 + wc * 64 + bq * 16 + fg * 4;
        ncol[bq] = n < p.Cout ? n : p.Cout - 4;
        u32x2 bb = {0u, 0u};
        if (p.bias) bb = *reinterpret_cast<const u32x2*>(p.bias + ncol[bq]);
        bv[bq][0] = bf16lo(bb[0]); bv[bq][1] = bf16hi(bb[0]); bv[bq][2] = bf16lo(bb[1]); bv[bq][3] = bf16hi(bb[1]);
    }
    char* tile = lds + wid * 16384;                      // 128 rows x 128 B; 16-byte chunk c of row r sits at chunk c ^ (r & 7)
    const int rl = lane >> 3, ch = lane & 7;
    int nc8 = n0 + wc * 64 + ch * 8;
    const bool nc8_ok = nc8 < p.Cout;
    nc8 = nc8_ok ? nc8 : p.Cout - 8;
    if constexpr (RES) {                                 // residual fetched row-wise (8 rows x 128 B per instruction) into the tile
        u32x4 rr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int loc = l0 + wr * 128 + i * 8 + rl;
            loc = loc < HW ? loc : HW - 1;
            rr[i] = *reinterpret_cast<const u32x4*>(p.res + ((int64_t)tf * HW + loc) * p.Cout + nc8);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 8 + rl;
            *reinterpret_cast<u32x4*>(tile + row * 128 + ((ch ^ (row & 7)) << 4)) = rr[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            const int row = a * 16 + fi, chunk = bq * 2 + (fg >> 1);
            char* at = tile + row * 128 + ((chunk ^ (row & 7)) << 4) + ((fg & 1) << 3);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[a][bq][j] + bv[bq][j];
            if constexpr (RES) {
                const u32x2 rv = *reinterpret_cast<const u32x2*>(at);
                v[0] += bf16lo(rv[0]); v[1] += bf16hi(rv[0]); v[2] += bf16lo(rv[1]); v[3] += bf16hi(rv[1]);
            }
            u32x2 o;
            o[0] = pack_bf16(v[0], v[1]);
            o[1] = pack_bf16(v[2], v[3]);
            *reinterpret_cast<u32x2*>(at) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = i * 8 + rl;
        const u32x4 val = *reinterpret_cast<const u32x4*>(tile + row * 128 + ((ch ^ (row & 7)) << 4));
        const int loc = l0 + wr * 128 + row;
        if (loc < HW && nc8_ok) *reinterpret_cast<u32x4*>(p.y + (m0 + wr * 128 + row) * p.Cout + nc8) = val;
    }
}

template <int WM, int WN, bool RES>
int launch_conv(const ConvGemmParams& p, hipStream_t st) {
    constexpr int LDS_BYTES = 2 * (2 * WM * 64 * 128 + 2 * WN * 32 * 128);
    static TcxPerDeviceOnce lds_attr;
    const int rc = tcx_ensure_dynamic_lds(lds_attr, reinterpret_cast<const void*>(&conv_mfma_kernel<WM, WN, RES>), LDS_BYTES, "tcx_conv3d_cl");
    if (rc != TCX_OK) return rc;
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RES>), dim3((unsigned)(p.mt * p.nt)), dim3(512), LDS_BYTES, st, p);
    TCX_LAUNCH_RET();
}

}  // namespace

// Shapes this kernel takes (everything else stays on conv.hip's register-staged kernel).
bool tcx_conv_mfma_supported(const TcxConvArgs& a) {
    const int64_t ktot = (int64_t)a.kT * a.kH * a.kW * a.Cin;
    return a.Cin % 64 == 0 && a.Cin <= 1024 && (ktot / BK) % 2 == 0 && a.Cout % 8 == 0 && a.Cout >= 128 &&
           (int64_t)(a.T_in + a.kT) * a.H_in * a.W_in * a.Cin < (1ll << 31);
}

int tcx_conv_mfma_launch(const TcxConvArgs& a, hipStream_t st) {
    ConvGemmParams p{};
    p.x = (const uint16_t*)a.x; p.cache = (const uint16_t*)a.cache; p.w = (const uint16_t*)a.w; p.bias = (const uint16_t*)a.bias;
    p.res = (const uint16_t*)a.res; p.y = (uint16_t*)a.y; p.t_map = a.t_map;
    p.T_in = a.T_in; p.W_in = a.W_in; p.Cin = a.Cin; p.Cout = a.Cout; p.kT = a.kT; p.kH = a.kH; p.kW = a.kW;
    p.T_out = a.T_out; p.H = a.H_out; p.W = a.W_out; p.ups = a.ups; p.LH = a.H_in << a.ups; p.LW = a.W_in << a.ups;
    p.stride = a.stride; p.pad_h = a.pad_h; p.pad_w = a.pad_w;
    p.Ktot = a.kT * a.kH * a.kW * a.Cin;
    p.M = (int64_t)a.N * a.T_out * a.H_out * a.W_out;
    p.frame_elems = (int64_t)a.H_in * a.W_in * a.Cin;
    const bool wide = a.Cout >= 256;                     // 256 x 256 tile; Cout == 128 (.. 255): 512 x 128
    const int BM = wide ? 256 : 512, BN = wide ? 256 : 128;
    p.frames = a.N * a.T_out;
    const int64_t mt = (int64_t)p.frames * (((int64_t)a.H_out * a.W_out + BM - 1) / BM);   // frames are tiled separately
    const int64_t nt = (a.Cout + BN - 1) / BN;
    TCX_CHECK(mt * nt < (1ll << 31) && p.M < (1ll << 40), TCX_E_SHAPE, "tcx_conv3d_cl: grid too large");
    p.mt = (int32_t)mt; p.nt = (int32_t)nt;
    if (wide) return a.res ? launch_conv<2, 4, true>(p, st) : launch_conv<2, 4, false>(p, st);
    return a.res ? launch_conv<4, 2, true>(p, st) : launch_conv<4, 2, false>(p, st);
}
