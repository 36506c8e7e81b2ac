// bf16 MFMA GEMM v3 for gfx950: persistent ping-pong kernel with 128-byte LDS rows (BK = 64) and 32x32x16 MFMAs.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )          same contract / epilogues as gemm_bf16.hip (v1) and v2
//
// What the v2 measurements said (rocprofv3 PMC + the timing ablation of tools/gemm_ablate.py, fc2 shape 147712x1024x4096):
//   full kernel 1.34 ms | MFMA + barriers only 0.85 ms | LDS-DMA stream + barriers only 0.93 ms | barriers only 0.33 ms
//   - the LDS-DMA stream alone is as slow as the math: v2's BK = 32 K-tiles make every global_load_lds fetch 64-byte
//     half lines (TCC requests = bytes / 64), and the L2 serves requests, not bytes: ~10 TB/s at this request size;
//   - every barrier interval carries ~150 cycles of fixed cost, v2 has one per 16 MFMAs (272 matrix-pipe cycles).
// v3 keeps what worked (256x256 tile, two groups of four waves skewed by one barrier so that one wave per SIMD is always in
// an MFMA segment, persistent blocks walking XCD-contiguous tile chunks, counted/explicit waits, hand-written fragment
// reads, hoisted epilogues) and changes the data path:
//   * K-tiles of 64 -> LDS rows of 128 B: every LDS-DMA instruction moves eight FULL 128-byte lines (1.5x the DMA rate,
//     measured below); slot ^= (row>>1)&7 keeps the 32-row fragment ds_read_b128 conflict free (same format as v1 / attention);
//   * an operand tile (256 rows x 64 k) is 32 KB; the whole 160 KB LDS is a ring of FIVE operand slots walked by the
//     item sequence X0 W0 X1 W1 X2 ...: while tile s is consumed, W(s+1) and X(s+2) stream in.  Each wave issues four
//     LDS-DMA loads per load segment (W quad in L0, X quad in L1), so the DMA bursts are balanced against the MFMA
//     segments and the urgent half (W of the next tile) is issued first; waits are counted, the queue never drains;
//   * per K-tile and group: L0 | M0 | L1 | M1 with 12 fragment reads per L and 16 v_mfma_f32_32x32x16_bf16 per M
//     (512 matrix-pipe cycles per barrier interval instead of 272).
//
// Measured DMA ceilings (tools/probes/dma_probe.hip, L2-resident panels): 64-byte row segments 21 TB/s = 34.5 B/clk/CU,
// 128-byte rows 31.6 TB/s = 51.5 B/clk/CU; a 256x256 tile needs 32 B/clk/CU at 100 % MFMA utilisation.
//
// Barrier / hazard ledger (s = stream index of a K-tile; item X(s) -> slot (2s) % 5, W(s) -> slot (2s+1) % 5;
// "instance" = global s_barrier count):
//   group 0:  L0(s) | b 4s+1 | M0(s) | b 4s+2 | L1(s) | b 4s+3 | M1(s) | b 4s+4
//   group 1:  (extra barrier = instance 1)  L0(s) | b 4s+2 | M0(s) | b 4s+3 | L1(s) | b 4s+4 | M1(s) | b 4s+5
//   issue:    both groups issue their W quad of tile s+1 in L0(s) and their X quad of tile s+2 in L1(s)
//             (prologue: X0, W0, X1).
//   WAR:      W(s+1) reuses the slot of X(s-1), X(s+2) the slot of W(s-1); tile s-1 was last read in L1(s-1): group 0's
//             reads retired before instance 4s-1, group 1's before instance 4s; the earliest overwrite is issued after
//             instance 4s (group 0's L0(s)).
//   RAW:      tile s+1 is first read after instance 4s+4.  Group 0 waits vmcnt(4) at the end of M1(s), group 1 at the
//             end of L1(s): only the X quad of tile s+2 (issued last) may still be in flight, so X(s+1) and W(s+1) have
//             landed on every wave before anyone passes instance 4s+4.
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

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int XW_BYTES = 256 * TK * 2;          // 32 KB per operand tile
constexpr int NSLOT = 5;
constexpr int LDS2 = NSLOT * XW_BYTES;            // 160 KB: all of the CU's LDS

VR_DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
VR_DEV void wait_vm4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
// Fragment reads are hand-written: hipcc's waitcnt pass cannot tell an LDS read from the bytes an in-flight LDS-DMA will
// write, so a ds_read it can see gets an s_waitcnt vmcnt(0) in front of it every K-tile, which drains the whole prefetch
// ring.  The loads and their lgkmcnt(0) live in ONE asm statement (early-clobber outputs), so no consumer and no
// register copy can be scheduled between issue and arrival; ordering against the DMA is the ledger above.
VR_DEV unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
VR_DEV void lds_issue12(bf16x8 (&x)[4][2], bf16x8 (&w)[2][2], unsigned xa0, unsigned xa1, unsigned wa0, unsigned wa1) {
    asm volatile(
        "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:4096\n\tds_read_b128 %2, %12 offset:8192\n\tds_read_b128 %3, %12 offset:12288\n\t"
        "ds_read_b128 %8, %14\n\tds_read_b128 %9, %14 offset:4096\n\t"
        "ds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:4096\n\tds_read_b128 %6, %13 offset:8192\n\tds_read_b128 %7, %13 offset:12288\n\t"
        "ds_read_b128 %10, %15\n\tds_read_b128 %11, %15 offset:4096"
        : "=&v"(x[0][0]), "=&v"(x[1][0]), "=&v"(x[2][0]), "=&v"(x[3][0]), "=&v"(x[0][1]), "=&v"(x[1][1]), "=&v"(x[2][1]), "=&v"(x[3][1]),
          "=&v"(w[0][0]), "=&v"(w[1][0]), "=&v"(w[0][1]), "=&v"(w[1][1])
        : "v"(xa0), "v"(xa1), "v"(wa0), "v"(wa1));
}
// the wait names every destination read-write: nothing that consumes (or copies) them can be scheduled above it
VR_DEV void lds_wait12(bf16x8 (&x)[4][2], bf16x8 (&w)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0][0]), "+v"(x[1][0]), "+v"(x[2][0]), "+v"(x[3][0]), "+v"(x[0][1]), "+v"(x[1][1]), "+v"(x[2][1]), "+v"(x[3][1]),
                   "+v"(w[0][0]), "+v"(w[1][0]), "+v"(w[0][1]), "+v"(w[1][1]));
}
VR_DEV void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

struct TileWalk {           // the block's list of output tiles: chunk of its XCD, strided by the blocks of that XCD
    int start, stride, count, ntn;
    VR_DEV void decode(int i, int& m0, int& n0) const {
        const int ii = i < count ? i : count - 1;        // past-the-end loads re-read the last tile (never consumed)
        const int t = start + ii * stride;
        m0 = (t / ntn) * TM;
        n0 = (t % ntn) * TN;
    }
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256p(const GemmArgs p) {
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
    }
    if (tw.count == 0) return;                                 // uniform per block: no barrier has been executed yet
    const int nk = p.K / TK;
    const int S = tw.count * nk;                               // K-tiles in this block's stream

    // ---- LDS-DMA source cursors.  Wave w covers rows [32w, 32w+32) of both operand tiles: 4 instructions of 8 rows x 128 B.
    //      lane -> (row = 8j + lane>>3, physical slot = lane&7); it fetches logical slot (lane&7) ^ ((row>>1)&7).
    struct Cur { const bf16_t* p[4]; int k, ti, idx; };
    Cur cx, cw;
    auto set_x = [&](Cur& c) {
        int m0, n0; tw.decode(c.ti, m0, n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int r = m0 + wave * 32 + j * 8 + (lane >> 3);
            r = r < p.M ? r : p.M - 1;                           // rows past M are computed but never stored
            c.p[j] = p.A + (size_t)r * p.lda + (((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) << 3);
        }
    };
    auto set_w = [&](Cur& c) {
        int m0, n0; tw.decode(c.ti, m0, n0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c.p[j] = p.W + (size_t)(n0 + wave * 32 + j * 8 + (lane >> 3)) * p.ldw + (((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) << 3);
    };
    cx.k = cw.k = 0; cx.ti = cw.ti = 0; cx.idx = cw.idx = 0;
    set_x(cx); set_w(cw);
    auto issue_x = [&]() {
        char* dst = smem + ((2 * cx.idx) % NSLOT) * XW_BYTES + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(cx.p[j] + cx.k, dst + j * 1024);
        ++cx.idx; cx.k += TK;
        if (cx.k == p.K) { cx.k = 0; ++cx.ti; set_x(cx); }
    };
    auto issue_w = [&]() {
        char* dst = smem + ((2 * cw.idx + 1) % NSLOT) * XW_BYTES + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(cw.p[j] + cw.k, dst + j * 1024);
        ++cw.idx; cw.k += TK;
        if (cw.k == p.K) { cw.k = 0; ++cw.ti; set_w(cw); }
    };

    // ---- fragment read offsets (32x32x16 operands): row = base32 + (lane&31), logical slot = 4*h + 2*kk + (lane>>5),
    //      physical slot = logical ^ ((row>>1)&7); the (h, kk) part only touches slot bits 1..2 -> one XOR per variant
    const int fr = lane & 31, hi = lane >> 5;
    const int fbase = fr * 128 + ((hi ^ ((fr >> 1) & 7)) << 4);
    const int xbase = grp * 128 * 128 + fbase;                 // + slot base + mi*4096 (immediate)
    const int wbase = wn * 64 * 128 + fbase;                   // + slot base + nj*4096 (immediate)

    // ---- prologue: X0, W0 landed and visible, X1 in flight
    issue_x(); issue_w(); issue_x();
    wait_vm4();
    barrier();

    const unsigned lds0 = lds_addr(smem);
    auto body = [&](auto G_) {
        constexpr int G = decltype(G_)::value;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{};
        int kt = 0, ti = 0;
        if (G == 1) barrier();                                 // skew: group 1 runs one barrier interval behind
        for (int s = 0; s < S; ++s) {
            const unsigned sx = lds0 + (unsigned)((2 * s) % NSLOT) * XW_BYTES, sw = lds0 + (unsigned)((2 * s + 1) % NSLOT) * XW_BYTES;
            bf16x8 xf[4][2], wf[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // ---------------- L(h): 12 fragment reads of k-half h + four LDS-DMA loads (L0: W of tile s+1, L1: X of tile s+2)
                const unsigned xa = (sx + xbase) ^ (h << 6), wa = (sw + wbase) ^ (h << 6);
                __builtin_amdgcn_s_setprio(1);                 // the load segment gets the issue priority
                if (!(DBG & 4)) lds_issue12(xf, wf, xa, xa ^ 32u, wa, wa ^ 32u);
                if (!(DBG & 2)) { if (h == 0) issue_w(); else issue_x(); }
                if (h == 1 && G == 1) wait_vm4();
                lds_wait12(xf, wf);
                __builtin_amdgcn_s_setprio(0);
                barrier();
                // ---------------- M(h): 16 MFMAs 32x32x16, nothing else
                if (!(DBG & 1))
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if (EPI == EPI_VT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[i][kk], wf[j][kk], acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][kk], xf[i][kk], acc[i][j], 0, 0, 0);
                        }
                if (h == 0) barrier();
            }
            __builtin_amdgcn_sched_barrier(0);                 // keep the counted wait behind the segment's MFMAs (hipcc hoists it otherwise)
            if (G == 0) wait_vm4();
            if (++kt == nk) {
                // ------------------------------------------------ epilogue of output tile ti
                kt = 0;
                int m0, n0; tw.decode(ti, m0, n0); ++ti;
                const int mb = m0 + grp * 128, nb = n0 + wn * 64;
                if (EPI == EPI_VT) gemm_epilogue_vt<4, 2>(p, acc, mb, nb, fr, hi);
                else gemm_epilogue_rowmajor<EPI, 4, 2>(p, acc, mb, nb, fr, hi);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{};
            }
            barrier();
        }
        if (G == 0) barrier();                                 // group 1 executed one extra barrier up front
    };
    if (grp == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    wait_vm0();                                                // drain the (unused) run-ahead loads before exit
}

template <int EPI>
int launch3(const GemmArgs& a, hipStream_t s) {
    static VisrepLdsOptIn opt;                                   // per (kernel instantiation, device)
    visrep_lds_opt_in(opt, reinterpret_cast<const void*>(gemm_bf16_256p<EPI>), LDS2);
    const int ncu = visrep_cu_count();
    const int ntiles = ((a.M + TM - 1) / TM) * (a.N / TN);
    const int grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL(gemm_bf16_256p<EPI>, dim3(grid), dim3(512), LDS2, s, a);
    return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
}

}  // namespace

// the patch-embed epilogue (float4 position rows on the 32x32 layout) does not fit the register budget here: v2 runs it
bool visrep_gemm_v3_supports(const GemmArgs& a) { return a.N % TN == 0 && a.K % TK == 0 && a.epi != EPI_PATCH; }

int visrep_gemm_v3_dispatch(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_BIAS: return launch3<EPI_BIAS>(a, s);
        case EPI_ACT: return launch3<EPI_ACT>(a, s);
        case EPI_RESID: return launch3<EPI_RESID>(a, s);
        case EPI_VT: return launch3<EPI_VT>(a, s);
        case EPI_F32: return launch3<EPI_F32>(a, s);
    }
    return visrep_set_error(VISREP_ERR_ARG, "gemm: unknown epilogue");
}
