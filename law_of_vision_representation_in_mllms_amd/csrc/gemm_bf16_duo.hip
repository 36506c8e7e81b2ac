// bf16 MFMA GEMM "duo" for gfx950 (round 6; gemm variants 6 / 7 / 8 - A/B variants, NOT the default: measured 20-30 % slower than v5 at every
// headline shape, profiles/round6_gemm.md section 2): TWO INDEPENDENT 4-wave workgroups per CU.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )          same contract / epilogues as gemm_bf16.hip (v1), v2 and v5; results BIT-IDENTICAL to v5's
//
// Why it exists.  The persistent 256x256 ping-pong kernel (v5) runs ONE 8-wave workgroup per CU: its two wave groups share
// the W ring and every s_barrier, so both finish an output tile together and neither epilogue overlaps any MFMA work - a tile boundary costs
// 12 % (fc2), 21 % (fc1), 35 % (out-proj) of the tile (profiles/round5_gemm.md section 2).  Round 5 argued the alternative away on LDS
// capacity without measuring it; this is that alternative, built to be measured: 256 x 128 output tiles, 4 waves (2 x 2, each 128 x 64 =
// the v5 wave tile: 8 x 4 accumulators of 16x16x32 MFMAs), a private LDS ring per workgroup and barriers of its own, two workgroups resident
// per CU (launch bounds 256 x 2, 72 KB of LDS each) - one workgroup's epilogue (and its pipeline fill, and its barrier waits) runs under the
// other's MFMAs because nothing couples them.  One output tile per workgroup (grid = tiles, XCD-contiguous order), no persistence.
//
// What the 80 KB per workgroup force.  A K-tile of 64 (v5's 128-byte LDS rows) is 48 KB for a 256 x 128 tile: not even two stages fit.  So the
// K-tile is 32 (64-byte LDS rows, v2's format): 16 KB of X + 8 KB of W = 24 KB per stage, a ring of THREE stages (72 KB).  The price is v2's:
// every LDS-DMA instruction moves sixteen 64-byte HALF lines (the L2 serves requests, not bytes), and a 256 x 128 tile stages 1.5x the operand
// bytes per FLOP of a 256 x 256 tile - which is what the measurement says decides it on a power-capped board.
//
// LDS layout: row r of an operand tile = 64 B = four 16-byte slots; slot s of row r sits at r * 64 + ((s ^ ((r >> 2) & 3)) << 4): a ds_read_b128
// of {16 consecutive rows, one logical slot} - lane l: row l & 15, slot l >> 4 - touches sixteen distinct 16-byte chunks of every 256-byte bank
// row (rows r .. r + 3 share a bank row and differ in (r & 3); rows r, r + 4, r + 8, r + 12 differ in the XOR term).  global_load_lds writes
// lane-linear (lane l -> byte 16 l of the 1-KB piece = row l >> 2, physical slot l & 3), so the lane fetches LOGICAL slot (l & 3) ^ ((row >> 2) & 3).
//
// Two schedules (template flag PIPE_, described at the kernel): the pipelined one (variants 6 / 7) overlaps a wave's fragment reads with its own
// MFMAs; the first build (variant 8) reads 12 fragments, waits, issues 32 MFMAs and leaves the gaps to the co-resident workgroup.  Both: one
// barrier per K-tile = per 512 matrix-pipe cycles (v5's cadence), counted vmcnt waits, hand-written ds_read_b128 (see gemm_bf16_v5.hip on why).
#include "common.h"
#include "gemm_epilogue.h"
#include "visrep_internal.h"

namespace {

constexpr int DM = 256, DN = 128, DK = 32;
constexpr int DX_BYTES = DM * DK * 2;            // 16 KB
constexpr int DW_BYTES = DN * DK * 2;            // 8 KB
constexpr int DSTAGE = DX_BYTES + DW_BYTES;      // 24 KB
constexpr int DNST = 3;
constexpr int DLDS = DNST * DSTAGE;              // 72 KB: two workgroups per CU

VR_DEV unsigned duo_lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
VR_DEV void duo_issue12(bf16x8 (&x)[8], bf16x8 (&w)[4], unsigned xa, unsigned wa) {
    asm volatile(
        "ds_read_b128 %8, %13\n\tds_read_b128 %9, %13 offset:1024\n\tds_read_b128 %10, %13 offset:2048\n\tds_read_b128 %11, %13 offset:3072\n\t"
        "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:1024\n\tds_read_b128 %2, %12 offset:2048\n\tds_read_b128 %3, %12 offset:3072\n\t"
        "ds_read_b128 %4, %12 offset:4096\n\tds_read_b128 %5, %12 offset:5120\n\tds_read_b128 %6, %12 offset:6144\n\tds_read_b128 %7, %12 offset:7168"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]),
          "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
        : "v"(xa), "v"(wa));
}
VR_DEV void duo_wait12(bf16x8 (&x)[8], bf16x8 (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
// pieces of the pipelined loop (PIPE_): the X fragments in two halves of four, W fragments on their own; waits are counted (LDS operations of a
// wave complete in order), and each wait names the registers it releases so that no consumer can be scheduled above it
VR_DEV void duo_issue4(bf16x8 (&f)[8], int o, unsigned a) {     // o = 0 | 4: which half (compile-time after unrolling)
    if (o == 0)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                     : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(a));
    else
        asm volatile("ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168"
                     : "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7]) : "v"(a));
}
VR_DEV void duo_issue_w(bf16x8 (&w)[4], unsigned a) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(a));
}
VR_DEV void duo_wait_lo(bf16x8 (&x)[8], bf16x8 (&w)[4]) {       // at most four reads (the upper X half) still in flight: x[0..3] and w have arrived
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
VR_DEV void duo_wait_hi(bf16x8 (&x)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
VR_DEV void duo_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// PIPE_ = true (variant 6 / 7): a wave's fragment reads run under its OWN MFMAs - the K-tile's 32 MFMAs in two halves of 16 (X rows 0-63 / 64-127 of
// the wave), the upper half's four X fragments are read under the lower half's MFMAs, the NEXT K-tile's lower half and W fragments (a second W
// set: +16 registers) under the upper half's; the stage wait + barrier sit between the halves, and because every wave has finished reading
// stage s by then, stage s + 3 goes into the slot stage s is leaving: three stages in flight behind the one being consumed.
// PIPE_ = false (variant 8, the first build): read 12 fragments, wait, 32 MFMAs - the co-resident workgroup alone has to fill the gaps.
template <int EPI, bool PIPE_>
__global__ __launch_bounds__(256, 2) void gemm_bf16_duo(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = p.N / DN, ntm = (p.M + DM - 1) / DM;
    const int t = xcd_remap(blockIdx.x, ntm * ntn);             // consecutive tiles on one XCD: 64 resident tiles = 8 row panels x all 8 column panels at N = 1024
    const int m0 = (t / ntn) * DM, n0 = (t % ntn) * DN;

    // ---- LDS-DMA sources.  X: 16 pieces of 16 rows, wave w stages pieces 4w .. 4w + 3; W: 8 pieces, wave w stages 2w, 2w + 1.
    const int pr = lane >> 2;                                    // row inside the piece
    const int lslot = (lane & 3) ^ ((pr >> 2) & 3);              // (piece bases are multiples of 16 rows: (row >> 2) & 3 == (pr >> 2) & 3)
    const bf16_t* gx[4];
    const bf16_t* gw[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int r = m0 + wave * 64 + q * 16 + pr;
        r = r < p.M ? r : p.M - 1;                               // rows past M are computed but never stored
        gx[q] = p.A + (size_t)visrep_a_row(p, r) * p.lda + lslot * 8;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) gw[q] = p.W + (size_t)(n0 + wave * 32 + q * 16 + pr) * p.ldw + lslot * 8;
    auto stage = [&](int buf, int kt) {
        char* sx = smem + buf * DSTAGE + wave * 4096;
        char* sw = smem + buf * DSTAGE + DX_BYTES + wave * 2048;
        const int ko = kt * DK;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(gx[q] + ko, sx + q * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(gw[q] + ko, sw + q * 1024);
    };

    // ---- fragment read offsets: row = 16 i + (lane & 15), logical slot = lane >> 4 (k = 8 (lane >> 4) .. + 8)
    const int fr = lane & 15, hi = lane >> 4;
    const unsigned fbase = (unsigned)(fr * 64 + ((hi ^ ((fr >> 2) & 3)) << 4));
    const unsigned lds0 = duo_lds_addr(smem);
    const unsigned xoff = lds0 + (unsigned)(wm * 128 * 64) + fbase;                 // + buf * DSTAGE (+ i * 1024 immediate)
    const unsigned woff = lds0 + (unsigned)(DX_BYTES + wn * 64 * 64) + fbase;       // + buf * DSTAGE (+ j * 1024 immediate)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / DK;
    auto mfma_half = [&](bf16x8 (&xf)[8], bf16x8 (&wf)[4], int o) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (EPI == EPI_VT) acc[o + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[o + i], wf[j], acc[o + i][j], 0, 0, 0);
                else acc[o + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[o + i], acc[o + i][j], 0, 0, 0);
            }
    };
    if constexpr (PIPE_) {
        // nk is even and >= 4 (visrep_gemm_duo_supports): two K-tiles per trip, the W sets alternate without a register copy
        stage(0, 0); stage(1, 1); stage(2, 2);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        duo_barrier();
        bf16x8 xf[8], wa[4], wb[4];
        duo_issue_w(wa, woff);
        duo_issue4(xf, 0, xoff);
        int buf = 0;
        auto ktile = [&](int kt, bf16x8 (&wcur)[4], bf16x8 (&wnext)[4]) __attribute__((always_inline)) {
            const unsigned so = (unsigned)(buf * DSTAGE);
            duo_issue4(xf, 4, xoff + so);                            // upper X half of stage kt
            duo_wait_lo(xf, wcur);
            mfma_half(xf, wcur, 0);
            __builtin_amdgcn_sched_barrier(0);
            duo_wait_hi(xf);                                         // every read of stage kt by this wave has completed
            const int nb = buf + 1 == DNST ? 0 : buf + 1;
            if (kt + 1 < nk) {
                if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // my pieces of stage kt + 1 have landed (stage kt + 2 may be in flight)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                duo_barrier();                                       // stage kt + 1 visible to all; stage kt read by all
                if (kt + 3 < nk) stage(buf, kt + 3);                 // into the slot stage kt leaves
                const unsigned sn = (unsigned)(nb * DSTAGE);
                duo_issue_w(wnext, woff + sn);
                duo_issue4(xf, 0, xoff + sn);                        // lower X half of stage kt + 1 (its registers are free since mfma_half(.., 0))
            }
            mfma_half(xf, wcur, 4);
            __builtin_amdgcn_sched_barrier(0);
            buf = nb;
        };
        for (int kt = 0; kt < nk; kt += 2) {
            ktile(kt, wa, wb);
            ktile(kt + 1, wb, wa);
        }
    } else {
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // my own pieces of stage kt have landed (the six of stage kt + 1 may stay in flight)
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        duo_barrier();                                           // stage kt visible to every wave; every wave has read stage kt - 1
        if (kt + 2 < nk) stage(buf >= 1 ? buf - 1 : DNST - 1, kt + 2);              // (kt + 2) % 3 == (kt - 1) % 3: the slot stage kt - 1 left
        bf16x8 xf[8], wf[4];
        duo_issue12(xf, wf, xoff + (unsigned)(buf * DSTAGE), woff + (unsigned)(buf * DSTAGE));
        duo_wait12(xf, wf);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (EPI == EPI_VT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], wf[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == DNST ? 0 : buf + 1;
    }
    }
    const int mb = m0 + wm * 128, nb = n0 + wn * 64;
    if constexpr (EPI == EPI_VT) gemm_epilogue_vt<8, 4>(p, acc, mb, nb, fr, hi);
    else gemm_epilogue_rowmajor<EPI, 8, 4, true>(p, acc, mb, nb, fr, hi);
}

template <int EPI, bool PIPE_>
int launch_duo(const GemmArgs& a, hipStream_t s) {
    static VisrepLdsOptIn opt;
    (void)visrep_lds_opt_in(opt, reinterpret_cast<const void*>(gemm_bf16_duo<EPI, PIPE_>), DLDS);
    const int ntiles = ((a.M + DM - 1) / DM) * (a.N / DN);
    hipLaunchKernelGGL((gemm_bf16_duo<EPI, PIPE_>), dim3(ntiles), dim3(256), DLDS, s, a);
    return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
}
template <bool PIPE_>
int dispatch_duo(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_BIAS: return launch_duo<EPI_BIAS, PIPE_>(a, s);
        case EPI_ACT: return launch_duo<EPI_ACT, PIPE_>(a, s);
        case EPI_RESID: return launch_duo<EPI_RESID, PIPE_>(a, s);
        case EPI_VT: return launch_duo<EPI_VT, PIPE_>(a, s);
    }
    return visrep_set_error(VISREP_ERR_ARG, "gemm duo: unsupported epilogue");
}

}  // namespace

bool visrep_gemm_duo_supports(const GemmArgs& a) {
    return !a.conv && a.N % DN == 0 && a.K % (2 * DK) == 0 && a.K >= 4 * DK && (a.epi == EPI_BIAS || a.epi == EPI_ACT || a.epi == EPI_RESID || a.epi == EPI_VT);
}

int visrep_gemm_duo_dispatch(const GemmArgs& a, hipStream_t s, bool pipelined) {
    return pipelined ? dispatch_duo<true>(a, s) : dispatch_duo<false>(a, s);
}
