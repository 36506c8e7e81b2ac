// bf16 MFMA GEMM v2 for gfx950: 256x256 tiles, 8 waves, ping-pong wave groups, 4-deep LDS ring, persistent blocks.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )          same contract / epilogues as gemm_bf16.hip (v1)
//
// Why: v1 (128x128, 4 waves, one barrier per K-tile, 2 blocks/CU) leaves the matrix pipe idle while a block's waves
// sit in their barrier + ds_read phase (measured 36 % of the bf16 peak at K = 4096, 21-31 % at K = 1024 where the
// pipeline fill and the epilogue are not amortised).  v2 restructures the block so that on every SIMD one wave is
// ALWAYS in an MFMA segment while its partner wave loads:
//
//   * 512 threads = 8 waves = two groups of four (waves w and w+4 share a SIMD).  Group g owns rows [128g, 128g+128)
//     of the 256x256 tile, wave wn = w&3 owns 64 columns: 128x64 per wave = 8x4 MFMA 16x16x32 accumulators.
//   * BK = 32 K-tiles (one MFMA k-step), staged by global_load_lds into a ring of FOUR 32-KB stages (X tile 256x32 +
//     W tile 256x32).  Loads run 2-3 K-tiles ahead of the math and are retired with COUNTED s_waitcnt vmcnt(N) -
//     the queue is never drained in the steady state.
//   * Each K-tile is two phases per group: {L: ds_read fragments + issue 2 LDS-DMA loads | M: 16 MFMAs}, separated
//     by raw s_barriers.  Group 1 is skewed by one barrier, so its L segments coincide with group 0's M segments
//     and vice versa (matrix beside memory on every SIMD, s_setprio(1) on the MFMA segment).
//   * The block is persistent: it walks its list of output tiles and the load stream simply continues into the next
//     tile's first K-tiles, so there is no pipeline refill between output tiles; blocks of one XCD walk a contiguous
//     chunk of the tile space (shared A / W panels in that XCD's L2).
//   * LDS rows are 64 B (four 16-B slots); slot ^= (4 - ((row>>2)&3))&3 makes the fragment ds_read_b128 conflict
//     free; the swizzle is applied on the per-lane global source address of the LDS-DMA (its LDS image is linear).
//
// Barrier / hazard ledger (s = stream index of a K-tile, stage = s & 3; "instance" = global s_barrier count):
//   group 0:  L0(s) | b 4s+1 | M0(s) | b 4s+2 | L1(s) | b 4s+3 | M1(s) | b 4s+4
//   group 1:  (extra barrier = instance 1)  L0(s) | b 4s+2 | M0(s) | b 4s+3 | L1(s) | b 4s+4 | M1(s) | b 4s+5
//   L segments: issue the fragment ds_reads, issue this wave's two LDS-DMA loads under the read latency, retire the
//   reads (lgkmcnt(0)), barrier.  M segments: 16 back-to-back MFMAs, nothing else (an LDS-DMA issue costs the issuing
//   wave ~60 cycles, which would come straight out of the matrix pipe if it sat between MFMAs - measured).
//   issue:    L0 carries the 8 fragment reads of the K-tile, L1 only 4, so all four LDS-DMA loads of a wave ride in L1:
//             group 0: W-pair of tile s+2 and X-pair of tile s+3 in L1(s);  group 1: X- and W-pair of tile s+3 in L1(s)
//   RAW:      tile s+1 is first read after instance 4s+4.  Before arriving there group 0 waits vmcnt(6) at the end of
//             M1(s) (outstanding: tile s+2 = 4, X-pair of s+3 = 2) and group 1 waits vmcnt(8) in L1(s) (tiles s+2,
//             s+3) -> tile s+1 has landed.
//   WAR:      stage (s+3)&3 was last read in tile s-1 (latest: group 1's L1(s-1), retired before instance 4s); the
//             earliest overwrite is issued after instance 4s+1.
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"
#include "visrep_internal.h"

// timing-only ablation (tools/gemm_ablate.py builds a separate library with -DVISREP_GEMM_ABLATE; production code has DBG == 0)
#ifdef VISREP_GEMM_ABLATE
#define DBG (p.dbg)
#else
#define DBG 0
#endif

#ifdef VISREP_GEMM_ABLATE
#define TSTAMP(i) do { if (timing) { const unsigned long long t_ = __builtin_readcyclecounter(); tsum[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

// 1: two of a wave's four LDS-DMA loads per K-tile go out in L0, two in L1; 0: all four in L1 (round 1).  The texture-addresser queue takes
// ~180 cycles per piece when 16 pieces arrive in one interval (tools/gemm_timing.py: L1 = 967 cycles with its four pieces, 236 without).
#ifndef VISREP_V2_DMA_SPLIT
#define VISREP_V2_DMA_SPLIT 1
#endif

namespace {

constexpr int TM = 256, TN = 256, TK = 32;
constexpr int XW_BYTES = 256 * TK * 2;          // 16 KB per operand tile
constexpr int STAGE2 = 2 * XW_BYTES;            // 32 KB
constexpr int NSTAGE = 4;                      // 128 KB ring (5 stages = all 160 KB measured identical: not latency-bound)
constexpr int LDS2 = NSTAGE * STAGE2;           // 128 KB

VR_DEV void wait_g0() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + 4 * (NSTAGE - 4)) : "memory"); }
VR_DEV void wait_g1() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + 4 * (NSTAGE - 4)) : "memory"); }
VR_DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
VR_DEV void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Fragment reads are hand-written: hipcc's waitcnt pass cannot tell an LDS read from the bytes an in-flight LDS-DMA will
// write, so a ds_read it can see gets an s_waitcnt vmcnt(0) in front of it every K-tile, which drains the whole prefetch
// ring.  The loads and their lgkmcnt(0) live in ONE asm statement (early-clobber outputs), so no consumer and no
// register copy can be scheduled between issue and arrival; ordering against the DMA is the ledger above.
VR_DEV unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
VR_DEV void lds_issue8(bf16x8 (&a)[4], bf16x8 (&b)[4], unsigned aaddr, unsigned baddr) {
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
        "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:1024\n\tds_read_b128 %6, %9 offset:2048\n\tds_read_b128 %7, %9 offset:3072"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
        : "v"(aaddr), "v"(baddr));
}
VR_DEV void lds_issue4(bf16x8 (&a)[4], unsigned aaddr) {
    asm volatile(
        "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
        : "v"(aaddr));
}
// the wait names every destination read-write: nothing that consumes (or copies) them can be scheduled above it
VR_DEV void lds_wait8(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
VR_DEV void lds_wait4(bf16x8 (&a)[4]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
VR_DEV void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

struct TileWalk {           // the block's list of output tiles: chunk of its XCD, strided by the blocks of that XCD
    int start, stride, count, ntn, ntm;
    // Tile order.  The ~32 blocks of an XCD work on ~32 CONSECUTIVE tile indices at any moment and share that XCD's L2, so the
    // set of operand panels behind a window of 32 indices should be small.  Row-major order gives 2 A panels + 16 W panels for
    // N = 4096 (fc1: W is re-fetched for every pair of row panels - 2.3 GB of its 2.85 GB HBM reads, profiles/round1_traffic.md);
    // walking R x 8 blocks (R = 4 row panels x 8 column panels) needs 4 + 8.  For ntn <= 8 both orders coincide.
    VR_DEV void decode(int i, int& m0, int& n0) const {
        const int ii = i < count ? i : count - 1;        // past-the-end loads re-read the last tile (never consumed)
        const int t = start + ii * stride;
        if (ntn > 8 && (ntn & 7) == 0) {
            const int R = 4, c = 8;
            const int sr = t / (R * ntn), u = t - sr * R * ntn;
            const int rl = min(R, ntm - sr * R);          // the last super-row may be shorter
            const int cg = u / (rl * c), v = u - cg * rl * c;
            m0 = (sr * R + v / c) * TM;
            n0 = (cg * c + v % c) * TN;
        } else {
            m0 = (t / ntn) * TM;
            n0 = (t % ntn) * TN;
        }
    }
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int ntn = p.N / TN, ntm = (p.M + TM - 1) / TM, ntiles = ntm * ntn;

    // ---- persistent tile list (XCD-contiguous chunks)
    TileWalk tw;
    {
        const int G = gridDim.x;
        const int nx = G < 8 ? G : 8;                          // XCDs in use
        const int x = blockIdx.x % nx, j = blockIdx.x / nx;   // block b runs on XCD b % 8 (speed only)
        const int per = (G + nx - 1 - x) / nx;                // blocks on this XCD
        const int q = ntiles / nx, r = ntiles % nx;
        const int cstart = x * q + (x < r ? x : r), csize = q + (x < r ? 1 : 0);
        tw.start = cstart + j;
        tw.stride = per;
        tw.count = j < csize ? (csize - j + per - 1) / per : 0;
        tw.ntn = ntn;
        tw.ntm = ntm;
    }
    if (tw.count == 0) return;                                 // uniform per block: no barrier has been executed yet
    const int nk = p.K / TK;
    const int S = tw.count * nk;                               // K-tiles in this block's stream

    // ---- LDS-DMA source cursors.  Wave w covers rows [32w, 32w+32) of both operand tiles: 2 instructions of 16 rows.
    const int lrow = lane >> 2;
    const int lslot = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    struct Cur { const bf16_t* p0; const bf16_t* p1; int k, ti, idx; };
    Cur cx, cw;
    auto set_x = [&](Cur& c) {
        int m0, n0; tw.decode(c.ti, m0, n0);
        int r0 = m0 + wave * 32 + lrow, r1 = r0 + 16;
        r0 = r0 < p.M ? r0 : p.M - 1; r1 = r1 < p.M ? r1 : p.M - 1;
        c.p0 = p.A + (size_t)visrep_a_row(p, r0) * p.lda + lslot * 8; c.p1 = p.A + (size_t)visrep_a_row(p, r1) * p.lda + lslot * 8;
    };
    auto set_w = [&](Cur& c) {
        int m0, n0; tw.decode(c.ti, m0, n0);
        const int r0 = n0 + wave * 32 + lrow;
        c.p0 = p.W + (size_t)r0 * p.ldw + lslot * 8; c.p1 = c.p0 + (size_t)16 * p.ldw;
    };
    cx.k = cw.k = 0; cx.ti = cw.ti = 0; cx.idx = cw.idx = 0;
    set_x(cx); set_w(cw);
    auto issue_x = [&]() {
        char* dst = smem + (cx.idx % NSTAGE) * STAGE2 + wave * 2048;
        glds16(cx.p0 + cx.k, dst); glds16(cx.p1 + cx.k, dst + 1024);
        ++cx.idx; cx.k += TK;
        if (cx.k == p.K) { cx.k = 0; ++cx.ti; set_x(cx); }
    };
    auto issue_w = [&]() {
        char* dst = smem + (cw.idx % NSTAGE) * STAGE2 + XW_BYTES + wave * 2048;
        glds16(cw.p0 + cw.k, dst); glds16(cw.p1 + cw.k, dst + 1024);
        ++cw.idx; cw.k += TK;
        if (cw.k == p.K) { cw.k = 0; ++cw.ti; set_w(cw); }
    };

    // ---- fragment read offset: row = base16 + (lane&15), logical slot = lane>>4
    const int fr = lane & 15, fg = lane >> 4;
    const int foff = fr * 64 + (((fg) ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4);
    const int xoff = grp * 128 * 64 + foff;                    // + mi*1024
    const int woff = XW_BYTES + wn * 64 * 64 + foff;           // + ni*1024

    // ---- prologue: tiles 0,1 (+2: group 1 fully, group 0 X-pair only) — leaves both groups in steady-state counts
#pragma unroll
    for (int t = 0; t < NSTAGE - 2; ++t) { issue_x(); issue_w(); }
    issue_x();
    if (grp == 1) { issue_w(); wait_g1(); } else { wait_g0(); }
    barrier();                                                 // tile 0 visible to every wave

    const unsigned lds0 = lds_addr(smem);
    auto body = [&](auto G_) __attribute__((always_inline)) {
        constexpr int G = decltype(G_)::value;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int kt = 0, ti = 0;
#ifdef VISREP_GEMM_ABLATE
        const bool timing = p.dbg_buf && blockIdx.x == 0 && wn == 0;
        unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
        if (G == 1) barrier();
#ifdef VISREP_GEMM_ABLATE
        tlast = __builtin_readcyclecounter();
#endif                                 // skew: group 1 runs one barrier interval behind
        for (int s = 0; s < S; ++s) {
            const unsigned sb = lds0 + (unsigned)(s % NSTAGE) * STAGE2;
            bf16x8 xf[4], wf[4];
            // ---------------- L0: W fragments (4) + X fragments of rows 0..63 (4); the wave's two LDS-DMA loads are issued
            //                  under the LDS read latency; everything is retired before the barrier
            if (!(DBG & 24)) __builtin_amdgcn_s_setprio(1);   // the load segment gets the issue priority (measured +20 % vs prio on the MFMA segment)
            if (!(DBG & 4)) lds_issue8(wf, xf, sb + woff, sb + xoff);
#if VISREP_V2_DMA_SPLIT
            if (!(DBG & 2)) { if (G == 0) issue_w(); else issue_x(); }   // first LDS-DMA pair of this K-tile's four: see L1
#endif
            lds_wait8(wf, xf);
            if (!(DBG & 24)) __builtin_amdgcn_s_setprio(0);
            TSTAMP(0);
            barrier();
            TSTAMP(1);                                         // fragments are in registers: M starts on the matrix pipe at once
            // ---------------- M0: 16 MFMAs, nothing else
            if (DBG & 16) __builtin_amdgcn_s_setprio(1);
            if (!(DBG & 1))
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (EPI == EPI_VT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], wf[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                }
            }
            if (DBG & 16) __builtin_amdgcn_s_setprio(0);
            TSTAMP(2);
            barrier();
            TSTAMP(3);
            // ---------------- L1: X fragments of rows 64..127 + this wave's other two LDS-DMA loads
            if (!(DBG & 24)) __builtin_amdgcn_s_setprio(1);
            if (!(DBG & 4)) lds_issue4(xf, sb + xoff + 4096);
#if VISREP_V2_DMA_SPLIT
            if (!(DBG & 2)) { if (G == 0) issue_x(); else issue_w(); }   // second pair (same piece ORDER as before, so the vmcnt ledger holds)
#else
            if (!(DBG & 2)) { if (G == 0) { issue_w(); issue_x(); } else { issue_x(); issue_w(); } }   // all four LDS-DMA loads ride in the short segment
#endif
            if (G == 1) wait_g1();
            lds_wait4(xf);
            if (!(DBG & 24)) __builtin_amdgcn_s_setprio(0);
            TSTAMP(4);
            barrier();
            TSTAMP(5);
            // ---------------- M1
            if (DBG & 16) __builtin_amdgcn_s_setprio(1);
            if (!(DBG & 1))
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (EPI == EPI_VT) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], wf[j], acc[4 + i][j], 0, 0, 0);
                    else acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[4 + i][j], 0, 0, 0);
                }
            }
            if (DBG & 16) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);                 // keep the counted wait behind the segment's MFMAs (hipcc hoists it otherwise)
            if (G == 0) wait_g0();
            TSTAMP(6);
            if (++kt == nk) {
                // ------------------------------------------------ epilogue of output tile ti
                kt = 0;
                int m0, n0; tw.decode(ti, m0, n0); ++ti;
                const int mb = m0 + grp * 128, nb = n0 + wn * 64;
                if (EPI == EPI_VT) gemm_epilogue_vt<8, 4>(p, acc, mb, nb, fr, fg);
                else gemm_epilogue_rowmajor<EPI, 8, 4, true>(p, acc, mb, nb, fr, fg);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            barrier();
            TSTAMP(7);
        }
        if (G == 0) barrier();                                 // group 1 executed one extra barrier up front
#ifdef VISREP_GEMM_ABLATE
        if (timing && lane == 0)
            for (int i = 0; i < 8; ++i) p.dbg_buf[G * 8 + i] = tsum[i];
#endif
    };
    if (grp == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    wait_vm0();                                                // drain the (unused) run-ahead loads before exit
}

template <int EPI>
int launch2(const GemmArgs& a, hipStream_t s) {
    static VisrepLdsOptIn opt;                                   // per (kernel instantiation, device)
    visrep_lds_opt_in(opt, reinterpret_cast<const void*>(gemm_bf16_256<EPI>), LDS2);
    const int ncu = visrep_cu_count();
    const int ntiles = ((a.M + TM - 1) / TM) * (a.N / TN);
    const int grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL(gemm_bf16_256<EPI>, dim3(grid), dim3(512), LDS2, s, a);
    return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
}

}  // namespace

bool visrep_gemm_v2_supports(const GemmArgs& a) { return a.N % TN == 0 && a.K % TK == 0; }

int visrep_gemm_v2_dispatch(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_BIAS: return launch2<EPI_BIAS>(a, s);
        case EPI_ACT: return launch2<EPI_ACT>(a, s);
        case EPI_RESID: return launch2<EPI_RESID>(a, s);
        case EPI_VT: return launch2<EPI_VT>(a, s);
        case EPI_PATCH: return launch2<EPI_PATCH>(a, s);
        case EPI_F32: return launch2<EPI_F32>(a, s);
    }
    return visrep_set_error(VISREP_ERR_ARG, "gemm: unknown epilogue");
}
