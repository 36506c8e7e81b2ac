// bf16 MFMA GEMM v5 for gfx950 (the default): persistent 256x256 ping-pong kernel, 128-byte LDS rows (BK = 64), 16x16x32 MFMAs.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )          same contract / epilogues as gemm_bf16.hip (v1) and v2
//
// Lineage.  v2 (256x256 tile, two groups of four waves skewed by one barrier so that one wave per SIMD is always in an MFMA
// segment, persistent XCD-contiguous tile walk, counted vmcnt waits, hand-written fragment reads) pays twice: its BK = 32 K-tiles
// make every LDS-DMA instruction fetch 64-byte half lines (the L2 serves requests, not bytes), and it has a barrier pair per 16
// MFMAs (~110 cycles of fixed cost per barrier interval).  v3 fixed both (BK = 64, five-slot ring, a barrier pair per 512
// matrix-pipe cycles) but computed with 32x32x16 MFMAs and lost the gain again.  v5 = v3's data path with v2's arithmetic:
//   * K-tiles of 64 -> LDS rows of 128 B: every LDS-DMA instruction moves eight FULL 128-byte lines; slot ^= (row>>1)&7 keeps the
//     fragment ds_read_b128 conflict free (same tile format as v1 / attention);
//   * an operand tile (256 rows x 64 k) is 32 KB; the whole 160 KB LDS is a ring of FIVE operand slots walked by the item
//     sequence X0 W0 X1 W1 X2 ...: while tile s is consumed, W(s+1) and X(s+2) stream in; waits are counted, the queue never drains;
//   * per K-tile and group: L0 | M0 | L1 | M1, one k-half (32 k) per L / M pair: an L segment is twelve ds_read_b128 (eight 16-row X
//     fragments of the group's 128 rows + four W fragments of the wave's 64 columns; 16-row steps leave (row>>1)&7 alone and the
//     k-half only flips slot bit 2, so every read is base ^ (h << 6) + immediate) plus the segment's LDS-DMA pieces; an M segment is
//     32 v_mfma_f32_16x16x32_bf16 (512 matrix-pipe cycles) on 8 x 4 accumulators and nothing else;
//   * epilogues are v2's (gemm_epilogue.h, 16x16 layout): 16-byte stores, LayerNorm fold, row statistics, patch embed.
//
// Who stages what (template flag OWN_, picked by N in the launcher; measurements in profiles/round2_gemm_v5.md):
//   OWN_ = false (v3's scheme): wave w stages rows [32w, 32w+32) of both operand tiles, four pieces each; both groups issue their
//     W quad of tile s+1 in L0(s) and their X quad of tile s+2 in L1(s).  Group 0 waits vmcnt(4) at the end of M1(s); group 1, whose
//     loads group 0 needs one barrier later, has to wait at the end of its L1(s) - inside a segment the other group's MFMAs wait behind.
//   OWN_ = true: group 0 stages ALL of W (wave wn: rows [64wn, 64wn+64), eight pieces in L0(s)) and its own upper half of X (four
//     pieces in L1(s)); group 1 stages only the lower half of X, which nobody but group 1 reads.  Both groups then wait vmcnt(4) at
//     the end of M1(s), under their own MFMAs.  Pays for wide outputs (fc1 +1.7 %, Q|K +2.4 %), costs 3 % at N = 1024 (8 pieces in one
//     load segment).
//
// Barrier / hazard ledger (s = stream index of a K-tile; item X(s) -> slot (2s) % 5, W(s) -> slot (2s+1) % 5; "instance" = global
// s_barrier count; prologue: X0, W0, X1):
//   group 0:  L0(s) | b 4s+1 | M0(s) | b 4s+2 | L1(s) | b 4s+3 | M1(s) | b 4s+4
//   group 1:  (extra barrier = instance 1)  L0(s) | b 4s+2 | M0(s) | b 4s+3 | L1(s) | b 4s+4 | M1(s) | b 4s+5
//   WAR:  W(s+1) reuses the slot of X(s-1), X(s+2) the slot of W(s-1); tile s-1 was last read in L1(s-1): group 0's reads retired
//         before instance 4s-1, group 1's before instance 4s; the earliest overwrite is issued after instance 4s (group 0's L0(s)).
//   RAW:  tile s+1 is first read after instance 4s+4 (group 0) / 4s+5 (group 1).  The counted wait leaves only the X quad of tile
//         s+2 (issued last) in flight.  OWN_ = false: every wave has confirmed W(s+1) and X(s+1) before instance 4s+4.  OWN_ = true:
//         group 0 has confirmed W(s+1) and X-upper(s+1) before instance 4s+4; X-lower(s+1) is confirmed by group 1 before instance
//         4s+5, which is all its only readers (group 1, from L0(s+1) on) need.
#include <type_traits>
#include <mutex>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "gemm_epilogue.h"
#include "visrep_internal.h"

// timing-only ablation (tools/gemm_ablate.py builds a separate library with -DVISREP_GEMM_ABLATE; production code has DBG == 0)
#ifdef VISREP_GEMM_ABLATE
#define DBG (p.dbg)
#else
#define DBG 0
#endif

// A/B knobs (tools/): LDS-DMA quad before or after the fragment reads of a load segment; s_setprio(1) on the load segments
#ifndef V5_DMA_FIRST
#define V5_DMA_FIRST 0
#endif
#ifndef V5_PRIO
#define V5_PRIO 1
#endif
// 1: group 0 (waves 0-3) stages ALL of W (64 rows = 8 pieces per wave and K-tile) and the upper half of X; group 1 stages only the lower half of
//    X - the half nobody but group 1 reads.  Group 1's counted wait then moves from the end of its load segment L1 (where it stalled a
//    segment the other group's MFMAs were waiting behind) to the end of its M1, under its own MFMAs, like group 0's.
// 0: every wave stages 32 rows of both operands (v3's scheme).
// Measured (same process, rows = 147,456): N >= 2048 gains (fc1 1.122 -> 1.103 ms, Q|K 0.544 -> 0.531), N = 1024 loses (fc2 0.938 -> 0.970, out 0.328 -> 0.337,
// V^T 0.269 -> 0.280): the kernel is compiled both ways and the launcher picks by N.  -DV5_OWN=0 / =1 forces one scheme (A/B builds).
#ifndef V5_DMA_IN_M
#define V5_DMA_IN_M 0
#endif
// A/B knobs (round 6, variant libraries): non-temporal cache policy (aux = 2) on the X / W LDS-DMA loads, so that the operand that is streamed
// once per round does not evict the one a column-group-major walk keeps in the XCD's L2 (profiles/round6_gemm.md)
#ifndef V5_NT_X
#define V5_NT_X 0
#endif
#ifndef V5_NT_W
#define V5_NT_W 0
#endif
#ifndef V5_OWN
#if V5_DMA_IN_M == 2
#define V5_OWN 0
#else
#define V5_OWN OWN_
#endif
#endif
// 1: group 0 meets the tile's last barrier BEFORE its epilogue instead of after it.  Both groups finish an output tile together (one barrier
//    interval apart), and with the barrier behind group 0's epilogue the two epilogues ran one after the other with the matrix pipe idle under
//    both: group 1 sat in that barrier (its M1 of the last K-tile waiting) for all of group 0's epilogue, then group 0 sat in the next one for
//    all of group 1's.  The epilogue touches no LDS and follows the group's counted wait, so nothing the barrier orders depends on it: with the
//    barrier in front, group 1's last 32 MFMAs and then its own epilogue run beside group 0's epilogue (two waves per SIMD in VALU / store
//    work instead of one), and the pipe idles for about one epilogue per tile instead of two (profiles/round4_gemm.md).
#ifndef V5_EPI_EARLY
#define V5_EPI_EARLY 1
#endif
// A/B knob V5_DMA_IN_M (round 4; measured slower, profiles/round4_gemm.md section 4): issue the LDS-DMA pieces BETWEEN the MFMAs of the M
// segments (one piece after every fourth MFMA) instead of inside the load segments, whose length - not the 512 matrix-pipe cycles of an M
// segment - sets the barrier cadence.  1: the OWN_ = true kernels only (group 0: 8 pieces in M0, 4 in M1; group 1: 4 in M1); the counted-wait
// ledger is unchanged (every piece is issued later than before, the waits stay put).  2: every kernel with the OWN_ = false staging (4 pieces
// per wave in M0 and in M1, both groups); group 1 then confirms W(s+1) and X(s+1) with vmcnt(0) at the end of L1(s) - nothing newer is in flight.

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int XW_BYTES = 256 * TK * 2;          // 32 KB per operand tile
constexpr int NSLOT = 5;
constexpr int LDS2 = NSLOT * XW_BYTES;            // 160 KB: all of the CU's LDS

template <int AUX>
VR_DEV void glds16a(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}
VR_DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
VR_DEV void wait_vm4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
// Fragment reads are hand-written: hipcc's waitcnt pass cannot tell an LDS read from the bytes an in-flight LDS-DMA will
// write, so a ds_read it can see gets an s_waitcnt vmcnt(0) in front of it every K-tile, which drains the whole prefetch
// ring.  The loads and their lgkmcnt(0) live in ONE asm statement (early-clobber outputs), so no consumer and no
// register copy can be scheduled between issue and arrival; ordering against the DMA is the ledger above.
VR_DEV unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
VR_DEV void lds_issue12(bf16x8 (&x)[8], bf16x8 (&w)[4], unsigned xa, unsigned wa) {     // W first: M starts with w[0..3] x x[0]
    asm volatile(
        "ds_read_b128 %8, %13\n\tds_read_b128 %9, %13 offset:2048\n\tds_read_b128 %10, %13 offset:4096\n\tds_read_b128 %11, %13 offset:6144\n\t"
        "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:2048\n\tds_read_b128 %2, %12 offset:4096\n\tds_read_b128 %3, %12 offset:6144\n\t"
        "ds_read_b128 %4, %12 offset:8192\n\tds_read_b128 %5, %12 offset:10240\n\tds_read_b128 %6, %12 offset:12288\n\tds_read_b128 %7, %12 offset:14336"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]),
          "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
        : "v"(xa), "v"(wa));
}
// the wait names every destination read-write: nothing that consumes (or copies) them can be scheduled above it
VR_DEV void lds_wait12(bf16x8 (&x)[8], bf16x8 (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
VR_DEV void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// End of a SAMPLED launch (GemmArgs::xb, one launch in 16): when did each XCD finish?  Every block leaves its end time in its XCD's maximum;
// the last block to arrive turns the eight maxima into "ticks per round of tiles" (a round = one tile on each of the XCD's blocks) and
// stores them in pinned host memory for visrep_xcd_plan, then zeroes the slot.  Off the hot path: after the last tile's stores.
VR_DEV void xcd_record(const GemmArgs& p, int ntiles, unsigned long long t0) {
    VisrepXcdSlot* s = &p.xb->slot[(p.xb_seq >> 3) & 15];
    const int G = gridDim.x, x = blockIdx.x & 7;
    atomicMax(&s->t_end[x], __builtin_amdgcn_s_memrealtime());
    if (blockIdx.x == 0) atomicExch(&s->t_start, t0);          // the blocks start within ~2 us of each other: block 0's start is the launch's
    __threadfence();
    if (atomicAdd(&s->done, 1u) != (unsigned)G - 1) return;
    __threadfence();
    const unsigned long long ts = atomicExch(&s->t_start, 0ull);
    const int per8 = G >> 3, q = ntiles >> 3, r = ntiles & 7;
    for (int y = 0; y < 8; ++y) {
        const unsigned long long te = atomicExch(&s->t_end[y], 0ull);
        const int n = p.xcd_bounds[8] == ntiles ? p.xcd_bounds[y + 1] - p.xcd_bounds[y] : q + (y < r ? 1 : 0);
        const int rounds = (n + per8 - 1) / per8;
        p.xb_host->tile_ticks[y] = (ts && te > ts && rounds > 0) ? (float)(te - ts) / (float)rounds : 0.f;
    }
    atomicExch(&s->done, 0u);
    __threadfence_system();
    p.xb_host->seq = p.xb_seq;
}

struct TileWalk {           // the block's list of output tiles: chunk of its XCD, strided by the blocks of that XCD
    int start, stride, count, ntn, ntm;
    int cgc, cstart, rows_x;    // column-group-major walk (round 6 experiment, GemmArgs::walk): cgc columns per group, the XCD's first tile, its row panels
    // Tile order as in v2: the ~32 blocks of an XCD work on ~32 consecutive tile indices and share that XCD's 4-MB L2; for more than 8
    // column panels (fc1: 16) the indices walk 4 x 8 blocks of tiles, so a window touches 4 + 8 operand panels instead of 2 + 16.
    VR_DEV void decode(int i, int& m0, int& n0) const {
        const int ii = i < count ? i : count - 1;        // past-the-end loads re-read the last tile (never consumed)
        const int t = start + ii * stride;
        if (cgc) {
            // The XCD owns rows_x whole row panels and walks them once per GROUP of cgc column panels: a round of its ~32 blocks is (32 / cgc) row
            // panels x cgc column panels, the next round the same columns and the next rows - the group's W panels (cgc x 512 KB at K = 1024) are what
            // stays in the XCD's 4-MB L2 across rounds, every X panel is fetched ntn / cgc times per launch instead of once per 4 x 8 window pair.
            const int u = t - cstart;
            const int cg = u / (rows_x * cgc), v = u - cg * rows_x * cgc;
            m0 = (cstart / ntn + v / cgc) * TM;
            n0 = (cg * cgc + v % cgc) * TN;
            return;
        }
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

__device__ __attribute__((aligned(16))) uint32_t g_zero_page5[4] = {0u, 0u, 0u, 0u};   // source of padded (out-of-image) conv taps

// CONV_ (round 4): implicit 3x3 convolution (NHWC, K index = tap * C + channel, C % 64 == 0 so a K-tile lies inside one tap; stride 1 | 2,
// symmetric or (0,1,0,1) padding, no upsampling): the X rows are output pixels, a lane's four row cursors point at the pixel's top-left tap
// and carry a 9-bit "tap is inside the image" mask; per K-tile the tap's offset (ky W + kx) C + c0 is one scalar and a lane whose tap is
// padding fetches the zero page instead.  W, the ring, the ledger and the epilogues are the plain GEMM's.
// GN_ (round 5, EPI_BIAS): the GroupNorm partial sums of the output (GemmArgs::gn_partial, as the 128x128 kernel's epilogue emits them) from a pass
// over the accumulators in front of the plain epilogue (gemm_epilogue.h gemm_gn_partials_prepass, which also says why not inside it), in an
// instantiation of its own.  Residual convolutions (round 6): the pre-pass reads their residual tile as well (RES), the epilogue reads it again.
template <int EPI, bool OWN_, bool CONV_ = false, bool GN_ = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256q(const GemmArgs p) {
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
        int cstart = x * q + (x < r ? x : r), csize = q + (x < r ? 1 : 0);
        if (p.xcd_bounds[8] == ntiles && nx == 8) {            // XCD-weighted shares (visrep_xcd_plan): faster XCDs walk longer chunks
            cstart = p.xcd_bounds[x];
            csize = p.xcd_bounds[x + 1] - cstart;
        }
        tw.start = cstart + j;
        tw.stride = per;
        tw.count = j < csize ? (csize - j + per - 1) / per : 0;
        tw.ntn = ntn;
        tw.ntm = ntm;
        const int wc = p.walk & 0xff;                          // column-group-major walk: whole row panels per XCD and whole groups, else the default order
        tw.cgc = (wc > 0 && ntn % wc == 0 && cstart % ntn == 0 && csize % ntn == 0 && csize > 0) ? wc : 0;
        tw.cstart = cstart;
        tw.rows_x = csize / ntn;
    }
    const unsigned long long xb_t0 = p.xb ? __builtin_amdgcn_s_memrealtime() : 0ull;      // sampled launches: when this block started (100-MHz ticks)
    if (tw.count == 0) { if (p.xb && tid == 0) xcd_record(p, ntiles, xb_t0); return; }    // uniform per block: no barrier has been executed yet
#ifdef V5_STAGGER_P                                             // A/B knob (tools/): block j of an XCD starts (j % P) x N sleeps of ~3.9 us late, so that the
    {                                                           // CUs' output bursts do not coincide (profiles/round4_gemm.md section 6)
        const int ph = (int)(blockIdx.x / 8) % V5_STAGGER_P;
        for (int i = 0; i < ph * V5_STAGGER_N; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int nk = p.K / TK;
    const int S = tw.count * nk;                               // K-tiles in this block's stream

    // ---- LDS-DMA source cursors.  Wave w covers rows [32w, 32w+32) of both operand tiles: 4 instructions of 8 rows x 128 B.
    //      lane -> (row = 8j + lane>>3, physical slot = lane&7); it fetches logical slot (lane&7) ^ ((row>>1)&7).
    // EPI_F32X: K = nprod * ksplit walks the (A plane, W plane) pairs of p.tab_a / p.tab_w (two bits per segment); the logical position k maps
    // to column plane * ksplit + (k mod ksplit) of the operand's [rows, nplanes * ksplit] plane matrix
    constexpr bool SPLIT = EPI == EPI_F32X;
    static_assert(!(CONV_ && (SPLIT || OWN_)), "the convolution gather exists for the plain staging scheme only");
    struct Cur { const bf16_t* p[4]; int k, ti, idx, kin, seg; int xo[4]; unsigned mk[2]; };     // xo / mk: CONV only (element offsets, 2 x 9-bit tap masks per word)
    Cur cx, cw;
    auto col_of = [&](const Cur& c, unsigned tab) { return SPLIT ? (int)((tab >> (2 * c.seg)) & 3u) * p.ksplit + c.kin : c.k; };
    const int seg_len = CONV_ ? p.cC : p.ksplit;              // CONV: seg = tap, kin = channel offset inside the tap
    auto advance = [&](Cur& c) {
        ++c.idx; c.k += TK;
        if (SPLIT || CONV_) { c.kin += TK; if (c.kin == seg_len) { c.kin = 0; ++c.seg; } }
    };
    auto set_x = [&](Cur& c) {
        int m0, n0; tw.decode(c.ti, m0, n0);
        if (CONV_) {
            // rows r0 + 8 j (j = 0..3): one division for the first, the others step 8 pixels along the output raster (cWo >= 8: one wrap at most).
            // Rows past M get an empty mask (zero page for every tap): their pixel may lie past the last image.
            const int r0 = m0 + wave * 32 + (lane >> 3);
            const int hw = p.cHo * p.cWo;
            const int g = r0 + p.a_row0;
            int b = g / hw;
            const int pix = g - b * hw;
            int oy = pix / p.cWo, ox = pix - oy * p.cWo;
            c.mk[0] = c.mk[1] = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy0 = oy * p.cstride - p.cpad, ix0 = ox * p.cstride - p.cpad;
                unsigned mk = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    mk |= (unsigned)((unsigned)(iy0 + t / 3) < (unsigned)p.cH && (unsigned)(ix0 + t % 3) < (unsigned)p.cW) << t;
                if (r0 + 8 * j >= p.M) mk = 0u;
                c.mk[j >> 1] |= mk << (16 * (j & 1));
                c.xo[j] = ((b * p.cH + iy0) * p.cW + ix0) * p.cC + ((((lane & 7) ^ ((4 * j + (lane >> 4)) & 7))) << 3);   // top-left tap (used only where the mask allows)
                ox += 8;
                if (ox >= p.cWo) { ox -= p.cWo; if (++oy == p.cHo) { oy = 0; ++b; } }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int r = m0 + wave * 32 + j * 8 + (lane >> 3);
            r = r < p.M ? r : p.M - 1;                           // rows past M are computed but never stored
            c.p[j] = p.A + (size_t)visrep_a_row(p, r) * p.lda + (((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) << 3);
        }
    };
    // the source of piece j for the K-tile the cursor stands on (CONV: tap = seg, channels kin .. kin + 64)
    auto x_src = [&](const Cur& c, int j, int kc) -> const bf16_t* {
        if (!CONV_) return c.p[j] + kc;
        const int ky = (c.seg >= 3) + (c.seg >= 6), kx = c.seg - 3 * ky;
        const int off = (ky * p.cW + kx) * p.cC + c.kin;       // uniform
        return ((c.mk[j >> 1] >> (c.seg + 16 * (j & 1))) & 1u) ? p.A + (ptrdiff_t)(c.xo[j] + off) : reinterpret_cast<const bf16_t*>(g_zero_page5);
    };
    auto set_w = [&](Cur& c) {
        int m0, n0; tw.decode(c.ti, m0, n0);
        if (V5_OWN) {   // wave w < 4 covers W rows [64w, 64w+64): piece j = row 8j + lane>>3; the swizzle term only depends on j & 1
#pragma unroll
            for (int j = 0; j < 2; ++j)
                c.p[j] = p.W + (size_t)(n0 + wn * 64 + j * 8 + (lane >> 3)) * p.ldw + (((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) << 3);
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c.p[j] = p.W + (size_t)(n0 + wave * 32 + j * 8 + (lane >> 3)) * p.ldw + (((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) << 3);
    };
    cx.k = cw.k = 0; cx.ti = cw.ti = 0; cx.idx = cw.idx = 0; cx.kin = cw.kin = 0; cx.seg = cw.seg = 0;
    set_x(cx); set_w(cw);
    auto issue_x = [&]() {
        char* dst = smem + ((2 * cx.idx) % NSLOT) * XW_BYTES + wave * 4096;
        const int kc = col_of(cx, p.tab_a);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16a<V5_NT_X ? 2 : 0>(x_src(cx, j, kc), dst + j * 1024);
        advance(cx);
        if (cx.k == p.K) { cx.k = 0; cx.kin = 0; cx.seg = 0; ++cx.ti; set_x(cx); }
    };
    auto issue_w = [&]() {
        const int kc = col_of(cw, p.tab_w);
        if (V5_OWN) {                                         // group 0 only: eight pieces, rows 64 wn + 8 j + lane>>3
            char* dst = smem + ((2 * cw.idx + 1) % NSLOT) * XW_BYTES + wn * 8192;
            const size_t step = (size_t)16 * p.ldw;
#pragma unroll
            for (int j = 0; j < 8; ++j) glds16a<V5_NT_W ? 2 : 0>(cw.p[j & 1] + (j >> 1) * step + kc, dst + j * 1024);
        } else {
            char* dst = smem + ((2 * cw.idx + 1) % NSLOT) * XW_BYTES + wave * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16a<V5_NT_W ? 2 : 0>(cw.p[j] + kc, dst + j * 1024);
        }
        advance(cw);
        if (cw.k == p.K) { cw.k = 0; cw.kin = 0; cw.seg = 0; ++cw.ti; set_w(cw); }
    };

    // piece-wise forms of issue_x / issue_w (V5_DMA_IN_M): piece j of the pending item, the cursor advances with the last piece
    auto issue_x_piece = [&](int j) {
        char* dst = smem + ((2 * cx.idx) % NSLOT) * XW_BYTES + wave * 4096;
        glds16(x_src(cx, j, col_of(cx, p.tab_a)), dst + j * 1024);
        if (j == 3) {
            advance(cx);
            if (cx.k == p.K) { cx.k = 0; cx.kin = 0; cx.seg = 0; ++cx.ti; set_x(cx); }
        }
    };
    auto issue_w_piece = [&](int j) {                         // OWN: group 0, eight pieces, rows 64 wn + 8 j + lane>>3; else four pieces like X
        if (!V5_OWN) {
            char* dst = smem + ((2 * cw.idx + 1) % NSLOT) * XW_BYTES + wave * 4096;
            glds16(cw.p[j] + col_of(cw, p.tab_w), dst + j * 1024);
            if (j == 3) {
                advance(cw);
                if (cw.k == p.K) { cw.k = 0; cw.kin = 0; cw.seg = 0; ++cw.ti; set_w(cw); }
            }
            return;
        }
        char* dst = smem + ((2 * cw.idx + 1) % NSLOT) * XW_BYTES + wn * 8192;
        const size_t step = (size_t)16 * p.ldw;
        glds16(cw.p[j & 1] + (j >> 1) * step + col_of(cw, p.tab_w), dst + j * 1024);
        if (j == 7) {
            advance(cw);
            if (cw.k == p.K) { cw.k = 0; cw.kin = 0; cw.seg = 0; ++cw.ti; set_w(cw); }
        }
    };

    // ---- fragment read offsets (16x16x32 operands): row = base16 + (lane&15), logical slot = 4*h + (lane>>4),
    //      physical slot = logical ^ ((row>>1)&7) (16-row steps leave (row>>1)&7 alone); the k-half h only flips slot bit 2 -> one XOR
    const int fr = lane & 15, hi = lane >> 4;
    const int fbase = fr * 128 + ((hi ^ ((fr >> 1) & 7)) << 4);
    const int xbase = grp * 128 * 128 + fbase;                 // + slot base + mi*2048 (immediate)
    const int wbase = wn * 64 * 128 + fbase;                   // + slot base + nj*2048 (immediate)

    // ---- prologue: X0, W0 landed and visible, X1 in flight
    issue_x();
    if (!V5_OWN || grp == 0) issue_w();
    issue_x();
    wait_vm4();
    barrier();

    const unsigned lds0 = lds_addr(smem);
    auto body = [&](auto G_) __attribute__((always_inline)) {
        constexpr int G = decltype(G_)::value;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int kt = 0, ti = 0;
#ifdef V5_TILE_TIMING                                       // diagnostic build (-DV5_TILE_TIMING, tools/gemm_tile_timing.py): cycle stamps at the tile boundaries of block 0
        const bool timing = p.dbg_buf && blockIdx.x == 0 && wn == 0;      // wave 0 (group 0) and wave 4 (group 1)
        unsigned long long t_loop = 0, t_epi = 0, t_mark = 0, t_bar = 0, t_b = 0;
        unsigned long long t_kb[5] = {0, 0, 0, 0, 0}, n_kb[5] = {0, 0, 0, 0, 0}, tk_prev = 0; int kt_prev = 0;
#endif
        if (G == 1) barrier();                                 // skew: group 1 runs one barrier interval behind
#ifdef V5_TILE_TIMING
        t_mark = __builtin_readcyclecounter();
        const unsigned long long t_first = t_mark, r_first = __builtin_amdgcn_s_memrealtime();   // shader-clock cycles / 100-MHz ticks: the clock this launch ran at
#endif
        for (int s = 0; s < S; ++s) {
            const unsigned sx = lds0 + (unsigned)((2 * s) % NSLOT) * XW_BYTES, sw = lds0 + (unsigned)((2 * s + 1) % NSLOT) * XW_BYTES;
            bf16x8 xf[8], wf[4];
#ifdef V5_TILE_TIMING                                       // whole K-tile iterations (closing barrier included) by their position in the output tile: 0, 1, 2,
            {                                                   // later ones, and the last one (which carries the epilogue and the closing barriers)
                const unsigned long long now = __builtin_readcyclecounter();
                if (s > 0) {
                    const unsigned long long d = now - tk_prev;
                    if (kt_prev == nk - 1) { t_kb[4] += d; ++n_kb[4]; } else if (kt_prev == 0) { t_kb[0] += d; ++n_kb[0]; } else if (kt_prev == 1) { t_kb[1] += d; ++n_kb[1]; }
                    else if (kt_prev == 2) { t_kb[2] += d; ++n_kb[2]; } else { t_kb[3] += d; ++n_kb[3]; }
                }
                tk_prev = now; kt_prev = kt;
            }
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // ---------------- L(h): 12 fragment reads of k-half h + four LDS-DMA loads (L0: W of tile s+1, L1: X of tile s+2)
                const unsigned xa = (sx + xbase) ^ (h << 6), wa = (sw + wbase) ^ (h << 6);
                if (V5_PRIO) __builtin_amdgcn_s_setprio(1);    // the load segment gets the issue priority
                constexpr bool IN_M = V5_DMA_IN_M == 2 || (V5_DMA_IN_M == 1 && OWN_);      // the pieces go out between the MFMAs of M(h) instead
                if (!IN_M && V5_DMA_FIRST && !(DBG & 2)) { if (h == 0) { if (!V5_OWN || G == 0) issue_w(); } else issue_x(); }
                if (!(DBG & 4)) lds_issue12(xf, wf, xa, wa);
                if (!IN_M && !V5_DMA_FIRST && !(DBG & 2)) { if (h == 0) { if (!V5_OWN || G == 0) issue_w(); } else issue_x(); }
                if (!V5_OWN && h == 1 && G == 1) { if (IN_M) wait_vm0(); else wait_vm4(); }
                lds_wait12(xf, wf);
                if (V5_PRIO) __builtin_amdgcn_s_setprio(0);
                barrier();
                // ---------------- M(h): 32 MFMAs 16x16x32, nothing else
                if (!(DBG & 1))
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (EPI == EPI_VT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], wf[j], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                    }
                    if (IN_M && !(DBG & 2)) {                    // one piece behind every fourth MFMA: W(s+1) in M0 (group 0: 8 pieces), X(s+2) in M1 (4)
                        __builtin_amdgcn_sched_barrier(0);
                        if (h == 0) { if (V5_OWN ? G == 0 : i < 4) issue_w_piece(i); }
                        else if (i < 4) issue_x_piece(i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (h == 0) barrier();
            }
            // the wait must stay BEHIND the segment's MFMAs (hipcc otherwise hoists it to after the first one: the wave then sits in the wait
            // with one MFMA in the pipe instead of 32) - a scheduling fence pins it
            __builtin_amdgcn_sched_barrier(0);
            if (G == 0 || V5_OWN) wait_vm4();
            if (++kt == nk) {
                // ------------------------------------------------ epilogue of output tile ti
                kt = 0;
                int m0, n0; tw.decode(ti, m0, n0); ++ti;
                const int mb = m0 + grp * 128, nb = n0 + wn * 64;
#ifdef V5_TILE_TIMING
                { const unsigned long long t = __builtin_readcyclecounter(); t_loop += t - t_mark; t_mark = t; }      // K loop of this tile (incl. its waits)
#endif
                if (V5_EPI_EARLY && G == 0) barrier();          // this iteration's closing barrier, taken before the epilogue (see V5_EPI_EARLY)
#ifdef V5_TILE_TIMING
                { const unsigned long long t = __builtin_readcyclecounter(); t_bar += t - t_mark; }                   // wait at the early barrier, if this group takes it here
#endif
                if constexpr (EPI == EPI_VT) gemm_epilogue_vt<8, 4>(p, acc, mb, nb, fr, hi);
                else if constexpr (EPI == EPI_F32X) gemm_epilogue_f32x<8, 4>(p, acc, mb, nb, fr, hi);
                else {
                    if constexpr (GN_) gemm_gn_partials_prepass<8, 4, EPI == EPI_RESID>(p, acc, mb, nb, fr, hi);
                    gemm_epilogue_rowmajor<EPI, 8, 4, true>(p, acc, mb, nb, fr, hi);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef V5_TILE_TIMING
                t_b = __builtin_readcyclecounter();             // end of the epilogue body (loads, arithmetic, store ISSUE, accumulator reset)
#endif
                if (!(V5_EPI_EARLY && G == 0)) barrier();
#ifdef V5_TILE_TIMING
                { const unsigned long long t = __builtin_readcyclecounter(); t_epi += t - t_mark; t_bar += t - t_b; t_mark = t; }   // barrier(s) + epilogue of this tile
#endif
            } else {
                barrier();
            }
        }
        if (G == 0) barrier();                                 // group 1 executed one extra barrier up front
#ifdef V5_TILE_TIMING
        if (timing && lane == 0) { p.dbg_buf[G * 8 + 0] = t_loop; p.dbg_buf[G * 8 + 1] = t_epi; p.dbg_buf[G * 8 + 2] = (unsigned long long)tw.count; p.dbg_buf[G * 8 + 3] = t_bar;
                                   p.dbg_buf[G * 8 + 4] = __builtin_readcyclecounter() - t_first; p.dbg_buf[G * 8 + 5] = __builtin_amdgcn_s_memrealtime() - r_first; }
        if (timing && lane == 0) {                             // K-tile time by position, without the barrier that closes the K-tile
            for (int i = 0; i < 5; ++i) { p.dbg_buf[16 + 4096 + G * 16 + i] = t_kb[i]; p.dbg_buf[16 + 4096 + G * 16 + 8 + i] = n_kb[i]; }
        }
        if (p.dbg_buf && G == 0 && wn == 0 && lane == 0) {     // every block: when its tile loop started / ended (100-MHz ticks, one clock for the whole device) and its cycles
            unsigned long long* o = p.dbg_buf + 16 + (size_t)blockIdx.x * 4;
            o[0] = r_first; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = __builtin_readcyclecounter() - t_first; o[3] = (unsigned long long)tw.count;
        }
#endif
    };
    if (grp == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    wait_vm0();                                                // drain the (unused) run-ahead loads before exit
    if (p.xb && tid == 0) xcd_record(p, ntiles, xb_t0);
}

template <int EPI, bool OWN_, bool CONV_ = false, bool GN_ = false>
int launch5o(const GemmArgs& a, hipStream_t s) {
    static VisrepLdsOptIn opt;                                   // per (kernel instantiation, device)
    visrep_lds_opt_in(opt, reinterpret_cast<const void*>(gemm_bf16_256q<EPI, OWN_, CONV_, GN_>), LDS2);
    const int ncu = visrep_cu_count();
    const int ntiles = ((a.M + TM - 1) / TM) * (a.N / TN);
    const int grid = ntiles < ncu ? ntiles : ncu;
    GemmArgs b = a;
    b.walk = t_visrep_gemm_walk;
    visrep_xcd_plan(b, s, grid, ntiles, reinterpret_cast<const void*>(gemm_bf16_256q<EPI, OWN_, CONV_, GN_>));
    hipLaunchKernelGGL((gemm_bf16_256q<EPI, OWN_, CONV_, GN_>), dim3(grid), dim3(512), LDS2, s, b);
    return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
}

template <int EPI>
int launch5(const GemmArgs& a, hipStream_t s) {
    return a.N >= 2048 ? launch5o<EPI, true>(a, s) : launch5o<EPI, false>(a, s);     // operand ownership pays for wide outputs only (see V5_OWN)
}

}  // namespace

// ------------------------------------------------------------------------------------------------ XCD-weighted tile split (host side)
namespace {
// One record per (kernel instantiation, M, N, K) and device: which XCD is slow depends on the kernel and its shape (operand placement in the
// HBM channels, L2 behaviour), not only on the XCD's clock - a device-wide estimate measured no gain (profiles/round5_gemm.md).
constexpr int XCD_RECORDS = 48;
struct XcdRecord {
    const void* fn = nullptr; int M = 0, N = 0, K = 0;
    float rel[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};   // smoothed time per round of tiles of each XCD, relative to the mean
    unsigned seen = 0, updates = 0, launches = 0;
};
struct XcdState {
    VisrepXcdDev* dev = nullptr;                                 // XCD_RECORDS of them
    VisrepXcdHost* host = nullptr;
    XcdRecord rec[XCD_RECORDS];
    int used = 0, last = -1;
    bool failed = false;
};
XcdState g_xcd[VISREP_MAX_DEVICES];
std::mutex g_xcd_mu;
std::atomic<int> g_xcd_on{-1};                                  // -1: not decided yet (VISREP_XCD_BALANCE), 0 / 1
bool xcd_enabled() {
    int v = g_xcd_on.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("VISREP_XCD_BALANCE");      // opt-in: measured +0.1 .. +0.3 % on the forward (profiles/round5_gemm.md section 3)
        v = (e && e[0] == '1') ? 1 : 0;
        g_xcd_on.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
}  // namespace

extern "C" int visrep_set_xcd_balance(int on) {
    const int old = xcd_enabled() ? 1 : 0;
    g_xcd_on.store(on ? 1 : 0, std::memory_order_relaxed);
    return old;
}
extern "C" int visrep_debug_xcd_balance(float* rel8, unsigned* updates) {     // the record of the most recent planned launch on this device
    const int dev = visrep_device();
    std::lock_guard<std::mutex> lk(g_xcd_mu);
    const XcdState& st = g_xcd[dev];
    const XcdRecord* r = st.last >= 0 ? &st.rec[st.last] : nullptr;
    if (rel8) for (int y = 0; y < 8; ++y) rel8[y] = r ? r->rel[y] : 1.f;
    if (updates) *updates = r ? r->updates : 0u;
    return xcd_enabled() ? 1 : 0;
}

// The split itself (pure host arithmetic, exported for tests): whole rounds (one tile on each of an XCD's grid / 8 blocks) in proportion to speed
// (rel8 = time per round of each XCD, any positive scale), the leftover rounds and the last partial round to whoever would finish first.
// bounds9[x] .. bounds9[x + 1] = XCD x's tile indices, bounds9[8] = ntiles.  Returns 0, or -1 for arguments the kernel does not take.
extern "C" int visrep_debug_xcd_split(const float* rel8, int grid, int ntiles, int* bounds9) {
    if (!rel8 || !bounds9 || grid < 8 || (grid & 7) || ntiles < grid) return -1;
    for (int y = 0; y < 8; ++y) if (!(rel8[y] > 0.f)) return -1;
    const int per8 = grid >> 3, rounds = ntiles / per8, rem = ntiles - rounds * per8;
    float inv = 0.f;
    for (int y = 0; y < 8; ++y) inv += 1.f / rel8[y];
    int R[8], tot = 0;
    for (int y = 0; y < 8; ++y) { R[y] = (int)((float)rounds / (rel8[y] * inv)); tot += R[y]; }
    for (; tot < rounds; ++tot) {
        int best = 0;
        for (int y = 1; y < 8; ++y) if ((R[y] + 1) * rel8[y] < (R[best] + 1) * rel8[best]) best = y;
        ++R[best];
    }
    for (; tot > rounds; --tot) {                                // (floating-point rounding of the floors above; not expected)
        int worst = 0;
        for (int y = 1; y < 8; ++y) if (R[y] * rel8[y] > R[worst] * rel8[worst]) worst = y;
        --R[worst];
    }
    int first = 0;
    for (int y = 1; y < 8; ++y) if (R[y] * rel8[y] < R[first] * rel8[first]) first = y;
    int acc = 0;
    for (int y = 0; y < 8; ++y) { bounds9[y] = acc; acc += R[y] * per8 + (y == first ? rem : 0); }
    bounds9[8] = acc;
    return 0;
}

void visrep_xcd_plan(GemmArgs& a, hipStream_t s, int grid, int ntiles, const void* fn) {
    a.xb = nullptr; a.xb_host = nullptr; a.xb_seq = 0; a.xcd_bounds[8] = 0;
    if (!xcd_enabled() || grid < 64 || (grid & 7) || ntiles < 8 * grid) return;     // a full chip and at least eight rounds of tiles per block
    // A launch that is being CAPTURED into a HIP graph runs with equal shares and is never a measurement: a plan baked into a graph would be
    // replayed forever with the bounds and the sample sequence number of capture time (the host would never see a new measurement and the
    // split would stop adapting - ADVICE r5).  The ViT and diffusion engines replay graphs: the opt-in split serves eager launches only.
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    }
    const int dev = visrep_device();
    std::lock_guard<std::mutex> lk(g_xcd_mu);
    XcdState& st = g_xcd[dev];
    if (st.failed) return;
    if (!st.dev) {                                               // first eligible launch on this device: allocate
        VisrepXcdDev* d = nullptr; VisrepXcdHost* h = nullptr;
        if (hipMalloc(&d, XCD_RECORDS * sizeof(VisrepXcdDev)) != hipSuccess || hipMemset(d, 0, XCD_RECORDS * sizeof(VisrepXcdDev)) != hipSuccess ||
            hipHostMalloc(&h, XCD_RECORDS * sizeof(VisrepXcdHost), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError(); st.failed = true; return;
        }
        memset(h, 0, XCD_RECORDS * sizeof(VisrepXcdHost));
        st.dev = d; st.host = h;
    }
    int k = -1;
    for (int i = 0; i < st.used; ++i)
        if (st.rec[i].fn == fn && st.rec[i].M == a.M && st.rec[i].N == a.N && st.rec[i].K == a.K) { k = i; break; }
    if (k < 0) {
        if (st.used == XCD_RECORDS) return;                      // table full: further shapes run with equal shares
        k = st.used++;
        st.rec[k].fn = fn; st.rec[k].M = a.M; st.rec[k].N = a.N; st.rec[k].K = a.K;
    }
    XcdRecord& r = st.rec[k];
    st.last = k;
    VisrepXcdHost* host = st.host + k;
    // a new measurement?  (plain reads of pinned memory the device writes: a stale or half-updated set of floats only delays the update)
    const unsigned seq = host->seq;
    if (seq != r.seen) {
        r.seen = seq;
        float t[8], mean = 0.f, lo = 1e30f, hi = 0.f;
        for (int y = 0; y < 8; ++y) { t[y] = host->tile_ticks[y]; mean += t[y] * 0.125f; lo = t[y] < lo ? t[y] : lo; hi = t[y] > hi ? t[y] : hi; }
        if (lo > 0.f && hi < 1.25f * lo) {                       // a clean sample: every XCD reported, spread within what clocks explain
            const float w = r.updates == 0 ? 1.f : 0.35f;        // the first sample replaces the prior, later ones are smoothed in
            for (int y = 0; y < 8; ++y) r.rel[y] = (1.f - w) * r.rel[y] + w * t[y] / mean;
            ++r.updates;
        }
    }
    visrep_debug_xcd_split(r.rel, grid, ntiles, a.xcd_bounds);
    const unsigned q = ++r.launches;
    if ((q & 7) == 4) { a.xb = st.dev + k; a.xb_host = host; a.xb_seq = q; }     // launches 4, 12, 20, ... of this record are measurements
}

bool visrep_gemm_v5_supports(const GemmArgs& a) {
    if (a.epi == EPI_F32X && (a.ksplit <= 0 || a.ksplit % TK || a.K % a.ksplit || a.K / a.ksplit < 1 || a.K / a.ksplit > 8)) return false;
    return a.N % TN == 0 && a.K % TK == 0;
}

// implicit 3x3 convolution in the 256x256 kernel: whole column tiles, 64-channel K-tiles inside one tap, no upsampled source
bool visrep_gemm_v5_supports_conv(const GemmArgs& a) {
    return a.conv && a.N % TN == 0 && a.cC % TK == 0 && a.K == 9 * a.cC && a.cup == 0 && a.kslice == 0 && a.cWo >= 8 && (a.epi == EPI_BIAS || a.epi == EPI_RESID) &&
           (long)a.M / (a.cHo * a.cWo) * a.cH * a.cW * a.cC < (1L << 31) - (3L * a.cW + 3) * a.cC;     // 32-bit element offsets
}

int visrep_gemm_v5_dispatch(const GemmArgs& a, hipStream_t s) {
    if (a.conv) {
        if (!visrep_gemm_v5_supports_conv(a)) return visrep_set_error(VISREP_ERR_ARG, "gemm v5: unsupported convolution");
        if (a.gn_partial) return a.epi == EPI_BIAS ? launch5o<EPI_BIAS, false, true, true>(a, s) : launch5o<EPI_RESID, false, true, true>(a, s);
        return a.epi == EPI_BIAS ? launch5o<EPI_BIAS, false, true>(a, s) : launch5o<EPI_RESID, false, true>(a, s);
    }
#ifndef V5_DEV_ONLY_CONV                                          // (development builds compile the convolution instantiations only)
    switch (a.epi) {
        case EPI_PATCH: return launch5<EPI_PATCH>(a, s);
        case EPI_BIAS: return launch5<EPI_BIAS>(a, s);
        case EPI_ACT: return launch5<EPI_ACT>(a, s);
        case EPI_RESID: return launch5<EPI_RESID>(a, s);
        case EPI_VT: return launch5<EPI_VT>(a, s);
        case EPI_F32: return launch5<EPI_F32>(a, s);
        case EPI_F32X: return launch5<EPI_F32X>(a, s);
    }
#endif
    return visrep_set_error(VISREP_ERR_ARG, "gemm: unknown epilogue");
}
