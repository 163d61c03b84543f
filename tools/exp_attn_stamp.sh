# In-kernel cycle stamps (s_memtime) of the shipped attention loop: loop total, barrier wait (and, for the kernels that still write
# their staged tiles in one segment — D = 128 and the coarse loops; the fine D = 64 loop spreads them over its steps since round 4 —
# that LDS-write segment), for waves 0 and 7 of two workgroups.  Patches a COPY of attn_fwd.hip; timing-only build.  Usage on the GPU box:
#   bash tools/exp_attn_stamp.sh
. "$(dirname "${BASH_SOURCE[0]}")/exp/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
cd $GRAFT_REPO_ROOT/trajectorycrafter_amd/csrc
python3 - <<'PY'
s = open("attn_fwd.hip").read()
# round 4: the staged tiles are written at the TOP of a super-step (D = 64) — stamp that segment and the barrier at its end
s = s.replace("""        if constexpr (WRITE_AT_TOP) write_staged();           // loaded a super-step ago
""", """        const long long tA = clock64();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WRITE_AT_TOP) write_staged();           // loaded a super-step ago
        __builtin_amdgcn_sched_barrier(0);
        st_write += clock64() - tA;
""")
s = s.replace("""#ifndef TCX_EXP_NOBARRIER
        __syncthreads();
#endif
    };""", """        __builtin_amdgcn_sched_barrier(0);
        const long long tB = clock64();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        st_bar += clock64() - tB;
        st_n += 1;
    };""")
s = s.replace("    auto run = [&](auto bnd) __attribute__((always_inline)) {", "    const long long tRun0 = clock64();\n    auto run = [&](auto bnd) __attribute__((always_inline)) {")
s = s.replace("    run(std::integral_constant<bool, BOUND>{});\n", """    run(std::integral_constant<bool, BOUND>{});
    if (BOUND && (blockIdx.x == 100 || blockIdx.x == 3001) && lane == 0 && (wave == 0 || wave == 7))
        printf("STAMP wg %d wave %d: loop %lld counts, %lld super-steps, write %lld, barrier %lld, tiles %d\\\\n", (int)blockIdx.x, wave,
               (long long)(clock64() - tRun0), st_n, st_write, st_bar, ntiles);
""")
s = s.replace("    bf16x8 kfa[2], kfb[2];\n", "    long long st_write = 0, st_bar = 0, st_n = 0;\n    bf16x8 kfa[2], kfb[2];\n", 1)
assert s.count("st_write") >= 3
open("/tmp/attn_stamp.hip", "w").write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. -x hip -c /tmp/attn_stamp.hip -o /tmp/attn_s.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_s.so tcx_api.o /tmp/attn_s.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o && \
TCX_LIB=/tmp/libtcx_s.so python3 $GRAFT_REPO_ROOT/tools/microbench.py attn --iters 1 2>&1 | grep "STAMP" | sort | uniq -c | sort -rn | head -8
