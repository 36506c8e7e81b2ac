// bf16 MFMA GEMM v4 for gfx950: 256x256 tile, FOUR waves (one per SIMD), 128x128 per wave, one software-pipelined instruction stream.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )          same contract / epilogues as gemm_bf16.hip (v1), v2 and v3
//
// Why (measured on v2 / v3, profiles/round1_pmc_gemm.md and DESIGN.md section 4.1): the 8-wave ping-pong kernels keep the matrix pipe
// ~50 % busy.  Their structure alternates, on every SIMD, one wave in a LOAD segment (ds_read fragments + LDS-DMA issue) with its
// partner in a 16-MFMA segment, four s_barriers per K-tile; the load segments (395 / 220 cycles) are longer than the MFMA segments
// (~315) and set the barrier cadence, and every wave re-reads (128 + 64) x K operand bytes from LDS for its 128x64 output.  v4 removes
// the alternation instead of tuning it:
//   * 256 threads = 4 waves, one per SIMD, each owning a 128x128 quadrant: 4x4 accumulators of v_mfma_f32_32x32x16_bf16 = 256
//     registers, held in the ACCUMULATOR half of the unified 512-entry register file ("a" operands); the 256 architectural VGPRs stay
//     free for two fragment sets, addresses and the epilogue.  Per 16-deep k-step a wave reads 4 + 4 fragments for 16 MFMAs
//     (0.5 ds_read_b128 per MFMA instead of 0.75, and half as many waves reading) and the matrix pipe never changes hands;
//   * ONE instruction stream per wave, hand-ordered (every MFMA, ds_read, LDS-DMA and wait is an `asm volatile` statement, which
//     hipcc keeps in program order): between the 16 MFMAs of k-step u sit the 8 fragment reads of k-step u+1 (into the other
//     fragment set) and 4 LDS-DMA pieces of a later K-tile - at 32 matrix-pipe cycles per MFMA each gap has room for ~5 issue slots
//     (MI355X_MICROARCH.md, "one wave per SIMD"), the stream uses 1-2;
//   * ONE s_barrier per 64-deep K-tile (64 MFMAs per wave, >= 2048 matrix-pipe cycles) instead of four per 32-deep tile.  It sits
//     between k-steps 2 and 3: by then every fragment of tile s is in registers (so the tile's two LDS slots are free for DMA issued
//     after it) and tile s+1 has landed (counted s_waitcnt vmcnt(8) in front of it), so k-step 3 already prefetches tile s+1's first
//     fragments - no pipeline bubble at the tile boundary;
//   * 128-byte LDS rows (BK = 64), slot ^= (row>>1)&7, five 32-KB operand slots = all 160 KB of LDS, item stream X0 W0 X1 W1 X2 ...
//     as in v3; the block is persistent and walks XCD-contiguous tile chunks (4x8 blocked for wide N) as in v2.
//
// Piece stream / hazard ledger.  A K-tile operand (256 rows x 128 B) is an ITEM of 32 LDS-DMA instructions (1 KB each: 8 rows x 128 B),
// 8 PIECES per wave.  Items are numbered q = 2 t (X of K-tile t) and 2 t + 1 (W of K-tile t) and live in slot q % 5.  Every wave issues
// its pieces in item order, 4 per k-step; the prologue issues 28 (items 0, 1, 2 and half of item 3).  With "barrier(t)" the barrier in
// K-tile t's body (after k-step 2):
//   pieces issued before barrier(t): 16 t + 40 -> items <= 2 t + 4 (all of X(t+2))
//   WAR  the 16 pieces issued between barrier(t-1) and barrier(t) belong to items 2 t + 3 (second half), 2 t + 4 and are written
//        into the slots of items 2 t - 2, 2 t - 1 = K-tile t - 1, whose last fragment read retired before barrier(t-1)
//        (s_waitcnt lgkmcnt(0) precedes every barrier).
//   RAW  K-tile t + 1 (items 2 t + 2, 2 t + 3 = pieces < 16 t + 32) is first read after barrier(t); each wave waits vmcnt(8) before
//        arriving: only its 8 newest pieces (item 2 t + 4) may still be in flight.  vmcnt also counts the epilogue's stores, which
//        only makes the wait more conservative.
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"
#include "visrep_internal.h"

// timing-only ablation (tools/gemm_v4_ablate.py builds libvisrep_hip_v4abl<mask>.so with -DV4_ABL=mask; production has 0):
// bit 0 no LDS-DMA, bit 1 no fragment reads, bit 2 no MFMAs - results are wrong for mask != 0
#ifndef V4_ABL
#define V4_ABL 0
#endif

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int XW_BYTES = 256 * TK * 2;          // 32 KB per operand tile
constexpr int NSLOT = 5;
constexpr int LDS4 = NSLOT * XW_BYTES;          // 160 KB: all of the CU's LDS

typedef f32x16 acc_t;

// ---- the instruction stream: everything is `asm volatile`, so program order is issue order
VR_DEV void mfma_acc(acc_t& c, const bf16x8& a, const bf16x8& b) {
    if (V4_ABL & 4) return;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
VR_DEV void mfma_zero(acc_t& c, const bf16x8& a, const bf16x8& b) {      // first k-step of an output tile: C = 0, the old value is dead
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
template <int OFF> VR_DEV void lds_read(bf16x8& f, unsigned addr) {
    if (V4_ABL & 2) return;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
// retire every outstanding LDS read; naming the fragments read-write keeps every consumer (and every register copy) below it
VR_DEV void lds_wait(bf16x8 (&x)[4], bf16x8 (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
// one LDS-DMA piece: 64 lanes x 16 B from (uniform 64-bit base + per-lane 32-bit byte offset) to LDS [dst, dst + 1 KB).
// M0 carries the wave-uniform LDS destination; it is compiler-reserved, so it is saved and restored inside the statement.
VR_DEV void dma_piece(unsigned voff, const void* sbase, unsigned lds_dst) {
    if (V4_ABL & 1) return;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
VR_DEV void wait_vm8() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
VR_DEV void wait_vm12() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
VR_DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
VR_DEV void hw_barrier() { asm volatile("s_barrier" ::: "memory"); }
VR_DEV void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }   // last MFMA's D -> first non-MFMA reader: 18 wait states
VR_DEV unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

struct TileWalk {           // the block's list of output tiles: chunk of its XCD, strided by the blocks of that XCD (as v2)
    int start, stride, count, ntn, ntm;
    VR_DEV void decode(int i, int& m0, int& n0) const {
        const int ii = i < count ? i : count - 1;        // past-the-end pieces re-read the last tile (never consumed)
        const int t = start + ii * stride;
        if (ntn > 8 && (ntn & 7) == 0) {                 // 4 x 8 blocked walk: 4 A panels + 8 W panels per XCD window (v2, traffic.md)
            const int R = 4, c = 8;
            const int sr = t / (R * ntn), u = t - sr * R * ntn;
            const int rl = min(R, ntm - sr * R);
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
__global__ __launch_bounds__(256) void gemm_bf16_256q(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(128))) char smem[];   // 128-B aligned: the k-step XOR on fragment addresses touches bits 5..6
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = p.N / TN, ntm = (p.M + TM - 1) / TM, ntiles = ntm * ntn;

    TileWalk tw;
    {
        const int G = gridDim.x;
        const int nx = G < 8 ? G : 8;
        const int x = blockIdx.x % nx, j = blockIdx.x / nx;
        const int per = (G + nx - 1 - x) / nx;
        const int q = ntiles / nx, r = ntiles % nx;
        const int cstart = x * q + (x < r ? x : r), csize = q + (x < r ? 1 : 0);
        tw.start = cstart + j;
        tw.stride = per;
        tw.count = j < csize ? (csize - j + per - 1) / per : 0;
        tw.ntn = ntn;
        tw.ntm = ntm;
    }
    if (tw.count == 0) return;
    const int nk = p.K / TK;
    const int S = tw.count * nk;                               // K-tiles in this block's stream

    // ---- LDS-DMA cursors.  Wave w moves rows [64 w, 64 w + 64) of every operand tile: 8 pieces of 8 rows x 128 B.
    //      lane -> (row = 8 j + lane>>3, physical 16-B slot = lane & 7), it fetches logical slot (lane & 7) ^ ((row >> 1) & 7)
    //      = (lane & 7) ^ ((4 j + (lane >> 4)) & 7).  Addresses are (uniform 64-bit base) + (per-lane 32-bit offset): the base carries
    //      the tile origin and the running k offset (scalar ALU), the per-lane offsets are constant per output tile.
    const unsigned lds0 = lds_addr(smem);
    struct Cur { unsigned off[8]; const char* base; int k, ti; };
    Cur cx, cw;
    auto set_x = [&](Cur& c) {
        int m0, n0; tw.decode(c.ti, m0, n0);
        const int rmax = p.M - 1 - m0;                           // rows past M are clamped (computed, never stored)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int r = wave * 64 + j * 8 + (lane >> 3);
            r = r < rmax ? r : rmax;
            c.off[j] = (unsigned)r * (unsigned)(p.lda * 2) + ((((lane & 7) ^ ((4 * j + (lane >> 4)) & 7))) << 4);
        }
        c.base = reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * 2;
    };
    auto set_w = [&](Cur& c) {
        int m0, n0; tw.decode(c.ti, m0, n0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            c.off[j] = (unsigned)(wave * 64 + j * 8 + (lane >> 3)) * (unsigned)(p.ldw * 2) + ((((lane & 7) ^ ((4 * j + (lane >> 4)) & 7))) << 4);
        c.base = reinterpret_cast<const char*>(p.W) + (size_t)n0 * p.ldw * 2;
    };
    cx.k = cw.k = 0; cx.ti = cw.ti = 0;
    set_x(cx); set_w(cw);
    int islot = 0;                                             // LDS slot of the item being issued (item % 5, kept incrementally)
    auto item_done = [&](auto ISW_) {                          // the running item's last piece went out: advance its operand's cursor
        constexpr bool ISW = decltype(ISW_)::value;
        Cur& c = ISW ? cw : cx;
        c.k += TK;
        if (c.k == p.K) { c.k = 0; ++c.ti; if (ISW) set_w(cw); else set_x(cx); }
        islot = islot == NSLOT - 1 ? 0 : islot + 1;
    };
    // prologue form: 4 pieces [J0, J0 + 4) of the running item back to back (X items are even, W items odd: static per call site)
    auto issue4 = [&](auto ISW_, auto J0_) {
        constexpr bool ISW = decltype(ISW_)::value;
        constexpr int J0 = decltype(J0_)::value;
        const Cur& c = ISW ? cw : cx;
        const unsigned dst = lds0 + (unsigned)islot * XW_BYTES + (unsigned)wave * 8192u;
        const char* b = c.base + (size_t)c.k * 2;
#pragma unroll
        for (int j = J0; j < J0 + 4; ++j) dma_piece(c.off[j], b, dst + j * 1024);
        if (J0 == 4) item_done(ISW_);
    };
    using I0 = std::integral_constant<int, 0>;
    using I4 = std::integral_constant<int, 4>;
    using OX = std::false_type;
    using OW = std::true_type;

    // ---- fragment read addresses: row = quadrant base + 32 i + (lane & 31), logical slot = 2 u + (lane >> 5) for k-step u,
    //      physical slot = logical ^ ((row >> 1) & 7); the k-step only touches slot bits 1..2 -> one XOR with u << 5 on the byte address
    const int fr = lane & 31, hi = lane >> 5;
    const unsigned fbase = (unsigned)(fr * 128 + ((hi ^ ((fr >> 1) & 7)) << 4));
    const unsigned xoff = (unsigned)(wm * 128 * 128) + fbase;   // + slot base, ^ (u << 5), + i * 4096 (immediate)
    const unsigned woff = (unsigned)(wn * 128 * 128) + fbase;

    // ---- prologue: items 0, 1, 2 and the first half of item 3 (28 pieces); K-tile 0 = the 16 oldest pieces
    issue4(OX{}, I0{}); issue4(OX{}, I4{}); issue4(OW{}, I0{}); issue4(OW{}, I4{}); issue4(OX{}, I0{}); issue4(OX{}, I4{}); issue4(OW{}, I0{});
    wait_vm12();
    __builtin_amdgcn_sched_barrier(0);
    hw_barrier();
    __builtin_amdgcn_sched_barrier(0);

    acc_t acc[4][4];
    bf16x8 x0[4], w0[4], x1[4], w1[4];                         // two fragment sets: k-steps 0 / 2 use set 0, k-steps 1 / 3 set 1
    auto read_set = [&](bf16x8 (&x)[4], bf16x8 (&w)[4], unsigned xa, unsigned wa, int n) {   // the n-th pair of the 8 reads of a set
        switch (n) {
            case 0: lds_read<0>(w[0], wa); lds_read<0>(x[0], xa); break;
            case 1: lds_read<4096>(w[1], wa); lds_read<4096>(x[1], xa); break;
            case 2: lds_read<8192>(w[2], wa); lds_read<8192>(x[2], xa); break;
            default: lds_read<12288>(w[3], wa); lds_read<12288>(x[3], xa); break;
        }
    };
    // one k-step: 16 MFMAs on (xc, wc); between them the 8 fragment reads of the NEXT k-step into (xn, wn) from addresses (xa, wa)
    // and the 4 LDS-DMA pieces [J0, J0 + 4) of the running item.  ZERO: first k-step of an output tile (accumulators start at 0).
    auto kstep = [&](auto ZERO_, auto ISW_, auto J0_, bf16x8 (&xc)[4], bf16x8 (&wc)[4], bf16x8 (&xn)[4], bf16x8 (&wnx)[4], unsigned xa, unsigned wa) {
        constexpr bool ZERO = decltype(ZERO_)::value;
        constexpr bool ISW = decltype(ISW_)::value;
        constexpr int J0 = decltype(J0_)::value;
        const Cur& c = ISW ? cw : cx;
        const unsigned dst = lds0 + (unsigned)islot * XW_BYTES + (unsigned)wave * 8192u;
        const char* b = c.base + (size_t)c.k * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = i * 4 + j;
                if (EPI == EPI_VT) { if (ZERO) mfma_zero(acc[i][j], xc[i], wc[j]); else mfma_acc(acc[i][j], xc[i], wc[j]); }
                else { if (ZERO) mfma_zero(acc[i][j], wc[j], xc[i]); else mfma_acc(acc[i][j], wc[j], xc[i]); }
                if (n < 8 && (n & 1) == 0) read_set(xn, wnx, xa, wa, n >> 1);        // after MFMAs 0, 2, 4, 6: two fragment reads each - the
                                                                                     // last one has 9 MFMAs (288 pipe cycles) to land
                if (n >= 8 && (n & 1) == 0) dma_piece(c.off[J0 + ((n - 8) >> 1)], b, dst + (J0 + ((n - 8) >> 1)) * 1024);   // after 8, 10, 12, 14: one DMA piece
            }
        }
        if (J0 == 4) item_done(ISW_);
    };
    using T = std::true_type;
    using F = std::false_type;

    // fragments of K-tile 0, k-step 0
    {
        const unsigned sx = lds0 + xoff, sw = lds0 + XW_BYTES + woff;
#pragma unroll
        for (int n = 0; n < 4; ++n) read_set(x0, w0, sx, sw, n);
        lds_wait(x0, w0);
    }
    int kt = 0, ti = 0;
    int tx = 0, twv = 1;                                       // LDS slots of X(s), W(s): (2 s) % 5 and (2 s + 1) % 5, kept incrementally
    for (int s = 0; s < S; ++s) {
        const int ntx = tx + 2 >= NSLOT ? tx + 2 - NSLOT : tx + 2, ntw = twv + 2 >= NSLOT ? twv + 2 - NSLOT : twv + 2;
        const unsigned sx = lds0 + (unsigned)tx * XW_BYTES + xoff, sw = lds0 + (unsigned)twv * XW_BYTES + woff;
        const unsigned nx = lds0 + (unsigned)ntx * XW_BYTES + xoff, nw = lds0 + (unsigned)ntw * XW_BYTES + woff;
        tx = ntx; twv = ntw;
        // k-step 0 (pieces 4..7 of item 2 s + 3 = second half of W(s+1)); reads k-step 1
        if (kt == 0) kstep(T{}, OW{}, I4{}, x0, w0, x1, w1, sx ^ 32u, sw ^ 32u);
        else kstep(F{}, OW{}, I4{}, x0, w0, x1, w1, sx ^ 32u, sw ^ 32u);
        lds_wait(x1, w1);
        // k-step 1 (pieces 0..3 of item 2 s + 4 = X(s+2)); reads k-step 2
        kstep(F{}, OX{}, I0{}, x1, w1, x0, w0, sx ^ 64u, sw ^ 64u);
        lds_wait(x0, w0);
        // k-step 2 (pieces 4..7 of X(s+2)); reads k-step 3 - the last fragments of K-tile s
        kstep(F{}, OX{}, I4{}, x0, w0, x1, w1, sx ^ 96u, sw ^ 96u);
        lds_wait(x1, w1);
        wait_vm8();                                            // K-tile s+1 has landed (this wave's share); X(s+2) may be in flight
        __builtin_amdgcn_sched_barrier(0);
        hw_barrier();                                          // barrier(s): K-tile s is in registers everywhere, K-tile s+1 is visible
        __builtin_amdgcn_sched_barrier(0);
        // k-step 3 (pieces 0..3 of item 2 s + 5 = W(s+2), into the slot K-tile s's X just left); reads k-step 0 of K-tile s+1
        kstep(F{}, OW{}, I0{}, x1, w1, x0, w0, nx, nw);
        lds_wait(x0, w0);
        if (++kt == nk) {
            kt = 0;
            mfma_drain();
            int m0, n0; tw.decode(ti, m0, n0); ++ti;
            const int mb = m0 + wm * 128, nb = n0 + wn * 128;
            // The accumulators live in the AGPR half of the register file; the shared epilogues want VGPR values.  One 32-row block
            // (4 accumulators = 64 registers) is copied over at a time - the empty asm statements pin each copy where it is written, so
            // hipcc cannot hoist all 256 copies above the epilogue (which spilled the fragment registers of the main loop).
            auto drain_rows = [&](auto I_) {                    // (a `for` over i is not unrolled by hipcc here: acc[i] would go to scratch)
                constexpr int i = decltype(I_)::value;
                acc_t t[1][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t[0][j] = acc[i][j];
                    asm volatile("" : "+v"(t[0][j]));
                }
                if (EPI == EPI_VT) gemm_epilogue_vt<1, 4>(p, t, mb + 32 * i, nb, fr, hi);
                else gemm_epilogue_rowmajor<EPI, 1, 4, true>(p, t, mb + 32 * i, nb, fr, hi);
            };
            drain_rows(std::integral_constant<int, 0>{});
            drain_rows(std::integral_constant<int, 1>{});
            drain_rows(std::integral_constant<int, 2>{});
            drain_rows(std::integral_constant<int, 3>{});
        }
    }
    wait_vm0();                                                // drain the (unused) run-ahead pieces before exit
}

template <int EPI>
int launch4(const GemmArgs& a, hipStream_t s) {
    static VisrepLdsOptIn opt;                                   // per (kernel instantiation, device)
    visrep_lds_opt_in(opt, reinterpret_cast<const void*>(gemm_bf16_256q<EPI>), LDS4);
    const int ncu = visrep_cu_count();
    const int ntiles = ((a.M + TM - 1) / TM) * (a.N / TN);
    const int grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL(gemm_bf16_256q<EPI>, dim3(grid), dim3(256), LDS4, s, a);
    return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
}

}  // namespace

// per-lane 32-bit byte offsets inside a 256-row operand panel: 255 * ld * 2 + 128 must fit (ld < 8 M elements - always)
bool visrep_gemm_v4_supports(const GemmArgs& a) { return a.N % TN == 0 && a.K % TK == 0 && a.epi != EPI_PATCH && !a.conv; }

int visrep_gemm_v4_dispatch(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_BIAS: return launch4<EPI_BIAS>(a, s);
        case EPI_ACT: return launch4<EPI_ACT>(a, s);
        case EPI_RESID: return launch4<EPI_RESID>(a, s);
        case EPI_VT: return launch4<EPI_VT>(a, s);
        case EPI_F32: return launch4<EPI_F32>(a, s);
    }
    return visrep_set_error(VISREP_ERR_ARG, "gemm: unknown epilogue");
}
