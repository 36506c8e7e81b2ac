// Head-width-64 attention forward, "two query blocks per wave" structure (variant 2 of visrep_attention_fwd; the ViT towers' MHSA).
//
// Why a second kernel: attn_fwd<1> (attention.hip) runs MFMA phase, softmax phase, MFMA phase strictly one after the other inside
// a wave (checked in the ISA) and leaves the overlap to the four waves per SIMD; PMC says the SIMD then executes ONE thing at a
// time (MFMA 30 %, softmax VALU 43 %).  At head width 64 the two kinds of work are the same size per key tile (16 MFMAs of 32
// cycles against ~140 VALU issue slots of ~4), so the kernel that reaches the roof must run them SIDE BY SIDE.  This one does it
// inside the wave:
//   * a wave owns TWO 32-row query blocks A and B (64 rows) and the whole 256-register budget (2 waves per SIMD, 2 blocks per CU);
//   * per key tile it runs two phases.  Phase 1: B's matrix work (S_B(t) = K(t) Q_B^T, then O_B += V^T(t-1) P_B(t-1)) interleaved,
//     MFMA by MFMA, with A's softmax of tile t (running max, rescale, exp2, row sum, bf16 pack).  Phase 2: A's matrix work
//     (S_A(t+1), then O_A += V^T(t) P_A(t)) interleaved with B's softmax of tile t.  Every MFMA has ~9-11 independent VALU
//     instructions of the OTHER query block behind it; the order is pinned with sched_barrier so hipcc cannot re-serialise it;
//   * the K / V^T fragment reads of MFMA i + 2 are issued in slot i; K and V^T tiles sit in two three-slot LDS rings filled by
//     LDS-DMA two tiles (K) / one tile (V^T) ahead, ONE barrier per key tile (per 32 MFMAs of a wave);
//   * the query blocks of an image-head are dealt out evenly over its workgroups (577 rows = 19 blocks -> 7 + 6 + 6), a wave
//     with one block runs the same phases without the partner stream, a wave with none only stages.
// Everything else (swapped S^T, in-register P, perm16 V^T layout, global 64-key tiles with masked edges, XCD remap, the
// lane-swap-widened stores) is attn_fwd<1>'s and is described there.
#include <type_traits>

#include "common.h"
#include "visrep_internal.h"

// 0: rescale O on every tile (alpha = 1 where a row's maximum did not move) - the textbook order, no branch in the stream.
// T > 0: keep the old maximum while no row's maximum grew by more than T (in exp2 units): P <= 2^T, the 32 multiplies per
//        block and tile leave the stream; the rare rescale runs in a cold branch.
#ifndef VISREP_ATTN_THR
#define VISREP_ATTN_THR 0
#endif
// timing-only: 1 = drop the sched_barrier pins (let hipcc order the phase)
#ifndef VISREP_ATTN_AB_NOPIN
#define VISREP_ATTN_AB_NOPIN 0
#endif

namespace {

constexpr int KT = 64;
constexpr int TILE_B = KT * 64 * 2;        // 8 KB
constexpr int NSLOT = 3;                   // ring depth of the K ring and of the V^T ring

struct AttnArgs2 {
    const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* out;
    int B, Tq, Tk, H, Mk, ldq, ldk, ldvt, ldo, kv_shared, causal;
    int nqb, nwg;                          // 32-row query blocks per image-head, workgroups per image-head
    float sc;
};

struct QBlk {
    f32x16 o[2];          // O^T accumulators (d tiles of 32)
    f32x16 s[2];          // raw scores of the tile in flight (two 32-key halves)
    uint32_t pb[2][8];    // bf16-packed P of the last softmax
    float m, l;           // running maximum (raw score units) and running sum of this lane's keys
    int qloc;             // this lane's query row inside the image
    const char* sq;       // this block's Q tile in LDS (32 rows x 128 B, swizzled like a K tile)
};

#if VISREP_ATTN_AB_NOPIN
#define PIN()
#else
#define PIN() __builtin_amdgcn_sched_barrier(0)
#endif

VR_DEV bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// Keys of X.s's tile outside [klo, klo + krange) (tile- and lane-relative, see mask_of) are set to -inf: edge tiles of an image and
// causal attention only, one contiguous in-place block in front of the phase.
VR_DEV void mask_scores(QBlk& X, int klo, int krange) {
    asm volatile("" : "+v"(klo), "+v"(krange));           // opaque: otherwise hipcc keeps 32 lane constants (c - klo) alive across the loop
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned c = kt2 * 32 + (r & 3) + 8 * (r >> 2);
            X.s[kt2][r] = (c - (unsigned)klo) < (unsigned)krange ? X.s[kt2][r] : -INFINITY;
        }
}

// One phase.  M side (matrix): DO_PV: M.o += V^T-tile(sv) P_M, then DO_QK: M.s = K-tile(sk) Q_M^T (Q fragments from M.sq).
// V side (softmax, DO_SM): V.s -> V.pb, V.m, V.l, V.o rescaled.
template <bool DO_QK, bool DO_PV, bool DO_SM>
VR_DEV void ab_phase(QBlk& M, QBlk& V, const char* sk, const char* sv, float sc, int rbase, int rsw, int hi) {
    constexpr int NM = (DO_QK ? 8 : 0) + (DO_PV ? 8 : 0);
    constexpr int NS = NM ? NM : 1;                       // slots
    constexpr int QK0 = DO_PV ? 8 : 0;                    // first Q.K^T op

    // ---- M side.  Matrix op order: the eight P.V MFMAs first (P dies as they go), then the eight Q.K^T ones (S is born late).
    bf16x8 fr[16], qfr[4];
    auto frag_addr = [&](int j) -> const char* {          // LDS operand of matrix op j
        const bool qk = DO_QK && j >= QK0;
        const int jj = qk ? j - QK0 : j;
        if (qk) { const int kk = jj >> 1, kt2 = jj & 1; return sk + kt2 * 4096 + rbase + (((2 * kk + hi) ^ rsw) << 4); }
        const int c = jj >> 1, dt = jj & 1;
        return sv + dt * 4096 + rbase + (((2 * c + hi) ^ rsw) << 4);
    };
    auto issue_reads = [&](int j) {                       // operand reads of matrix op j (the Q fragment with the first op that uses it)
        if (j >= NM) return;
        if (DO_QK && j >= QK0 && ((j - QK0) & 1) == 0) { const int kk = (j - QK0) >> 1; qfr[kk] = lds_frag(M.sq + rbase + (((2 * kk + hi) ^ rsw) << 4)); }
        fr[j] = lds_frag(frag_addr(j));
    };
    auto mat_op = [&](int j) {
        const bool qk = DO_QK && j >= QK0;
        const int jj = qk ? j - QK0 : j;
        if (qk) {
            const int kk = jj >> 1, kt2 = jj & 1;
            const f32x16 acc = kk == 0 ? f32x16{} : M.s[kt2];
            M.s[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j], qfr[kk], acc, 0, 0, 0);
        } else {
            const int c = jj >> 1, dt = jj & 1;
            u32x4 w = {M.pb[c >> 1][4 * (c & 1) + 0], M.pb[c >> 1][4 * (c & 1) + 1], M.pb[c >> 1][4 * (c & 1) + 2], M.pb[c >> 1][4 * (c & 1) + 3]};
            M.o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j], *reinterpret_cast<bf16x8*>(&w), M.o[dt], 0, 0, 0);
        }
    };

    // ---- V side: the softmax of V.s cut into slot-sized pieces
    float mloc0, mloc1, nmsc = 0.f, alpha = 1.f, ps0 = 0.f, ps1 = 0.f;
    auto sm_max_a = [&]() {
        mloc0 = fmaxf(fmaxf(V.s[0][0], V.s[0][1]), V.s[0][2]); mloc1 = fmaxf(fmaxf(V.s[1][0], V.s[1][1]), V.s[1][2]);
#pragma unroll
        for (int r = 3; r < 9; r += 2) { mloc0 = fmaxf(fmaxf(mloc0, V.s[0][r]), V.s[0][r + 1]); mloc1 = fmaxf(fmaxf(mloc1, V.s[1][r]), V.s[1][r + 1]); }
    };
    auto sm_max_b = [&]() {
#pragma unroll
        for (int r = 9; r < 15; r += 2) { mloc0 = fmaxf(fmaxf(mloc0, V.s[0][r]), V.s[0][r + 1]); mloc1 = fmaxf(fmaxf(mloc1, V.s[1][r]), V.s[1][r + 1]); }
        mloc0 = fmaxf(fmaxf(mloc0, V.s[0][15]), fmaxf(mloc1, V.s[1][15]));
    };
    auto sm_decide = [&]() {                              // new maximum; l moves to it here, O in the slots that follow (or in the cold branch)
        const unsigned u = __builtin_bit_cast(unsigned, mloc0);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        const float mloc = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
#if VISREP_ATTN_THR > 0
        if (__any((mloc - V.m) * sc > (float)VISREP_ATTN_THR)) {      // wave-uniform; V.m = -inf on the first tile -> taken
            const float m_new = fmaxf(V.m, mloc);
            alpha = __builtin_amdgcn_exp2f((V.m - m_new) * sc);
            V.m = m_new;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) V.o[dt][r] *= alpha;
        }
#else
        const float m_new = fmaxf(V.m, mloc);
        alpha = __builtin_amdgcn_exp2f((V.m - m_new) * sc);
        V.m = m_new;
#endif
        nmsc = -V.m * sc;
    };
    auto sm_rescale = [&](int e0, int e1) {               // O elements [e0, e1) of 32, in place and pinned (hipcc otherwise sinks all 32 multiplies
#if VISREP_ATTN_THR == 0                                  // to the head of the next phase, in front of its first MFMA)
#pragma unroll
        for (int e = e0; e < e1; ++e) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(V.o[e >> 4][e & 15]) : "v"(alpha));
#endif
    };
    // One asm statement per score pair: the seven instructions stay together and in program order (plain C++ here lets hipcc sink all
    // thirty-two row-sum adds to the end of the phase, which keeps every exponential alive - 32 registers - and un-does the
    // interleave).  v_exp is a transcendental: its consumer must not be the very next instruction (one independent one between).
    auto sm_pair = [&](int q) {                           // score elements 2q, 2q + 1 of 32
        const int kt2 = q >> 3, r = (q & 7) * 2;
        float a0, a1;
        asm volatile("v_fma_f32 %0, %5, %7, %8\n\t"
                     "v_fma_f32 %1, %6, %7, %8\n\t"
                     "v_exp_f32 %0, %0\n\t"
                     "v_exp_f32 %1, %1\n\t"
                     "v_add_f32 %2, %2, %0\n\t"
                     "v_add_f32 %3, %3, %1\n\t"
                     "v_cvt_pk_bf16_f32 %4, %0, %1"
                     : "=&v"(a0), "=&v"(a1), "+v"(ps0), "+v"(ps1), "=v"(V.pb[kt2][r >> 1])
                     : "v"(V.s[kt2][r]), "v"(V.s[kt2][r + 1]), "s"(sc), "v"(nmsc));
    };
    // Sixteen-slot plan: slot 0 carries no softmax work (the Q.K^T that produced V.s ended the previous phase: its result latency
    // passes under this phase's first MFMA), two slots of running maximum, one decision slot, sixteen score pairs over twelve slots.
    auto sm_slot16 = [&](int i) {
        if (i == 1) sm_max_a();
        else if (i == 2) sm_max_b();
        else if (i == 3) sm_decide();
        else if (i > 3) {
            const int k = i - 4;                          // 0 .. 11
            sm_rescale((k * 32) / 12, ((k + 1) * 32) / 12);
#pragma unroll
            for (int q = (k * 16) / 12; q < ((k + 1) * 16) / 12; ++q) sm_pair(q);
        }
    };

    issue_reads(0);
    issue_reads(1);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (NM) {
            issue_reads(i + 2);
            mat_op(i);
        }
        if (DO_SM) {
            if (NS == 16) sm_slot16(i);
            else if (NS == 8) { sm_slot16(2 * i); sm_slot16(2 * i + 1); }
            else {
#pragma unroll
                for (int j = 0; j < 16; ++j) sm_slot16(j);
            }
        }
        PIN();
    }
    if (DO_SM) V.l = __builtin_fmaf(V.l, alpha, ps0 + ps1);
}

// Q rows of one query block: global -> registers -> this block's swizzled LDS tile (read back as MFMA B fragments every tile).
VR_DEV void qblk_init(QBlk& X, const AttnArgs2& p, int b, int h, int qb, int lq, int hi, char* sq) {
    X.qloc = qb * 32 + lq;
    X.sq = sq;
    const size_t qrow = (size_t)b * p.Tq + (X.qloc < p.Tq ? X.qloc : p.Tq - 1);
    const int rsw = (lq >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        *reinterpret_cast<bf16x8*>(sq + lq * 128 + (((2 * kk + hi) ^ rsw) << 4)) = *reinterpret_cast<const bf16x8*>(p.q + qrow * p.ldq + h * 64 + kk * 16 + hi * 8);
    X.o[0] = f32x16{}; X.o[1] = f32x16{};
    X.m = -INFINITY; X.l = 0.f;
}

VR_DEV void qblk_store(QBlk& X, const AttnArgs2& p, int b, int h, int hi) {
    X.l += __shfl_xor(X.l, 32);
    const float inv = 1.f / X.l;
    bf16_t* orow = p.out + ((size_t)b * p.Tq + (X.qloc < p.Tq ? X.qloc : p.Tq - 1)) * p.ldo + h * 64;
    const bool wide = (p.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;          // uniform
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; rg += 2) {
            u32x2 a = {pack_bf16(X.o[dt][4 * rg + 0] * inv, X.o[dt][4 * rg + 1] * inv), pack_bf16(X.o[dt][4 * rg + 2] * inv, X.o[dt][4 * rg + 3] * inv)};
            u32x2 c = {pack_bf16(X.o[dt][4 * rg + 4] * inv, X.o[dt][4 * rg + 5] * inv), pack_bf16(X.o[dt][4 * rg + 6] * inv, X.o[dt][4 * rg + 7] * inv)};
            if (wide) {
                const auto w0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                const auto w1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                if (X.qloc < p.Tq) {
                    const u32x4 q4 = {(unsigned)w0[0], (unsigned)w1[0], (unsigned)w0[1], (unsigned)w1[1]};
                    *reinterpret_cast<u32x4*>(orow + dt * 32 + (rg + hi) * 8) = q4;
                }
            } else if (X.qloc < p.Tq) {
                *reinterpret_cast<u32x2*>(orow + dt * 32 + rg * 8 + hi * 4) = a;
                *reinterpret_cast<u32x2*>(orow + dt * 32 + (rg + 1) * 8 + hi * 4) = c;
            }
        }
}

__global__ __launch_bounds__(256, 2) void attn_fwd_ab(const AttnArgs2 p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // K ring: NSLOT x 8 KB, V^T ring: NSLOT x 8 KB, Q tiles: 8 x 4 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int wg = id % p.nwg; id /= p.nwg;
    const int h = id % p.H;
    const int b = id / p.H;
    const int tok0 = p.kv_shared ? 0 : b * p.Tk, tok1 = tok0 + p.Tk;

    // query blocks of this workgroup: an even contiguous split of the image-head's nqb blocks over its nwg workgroups
    const int qb0 = (wg * p.nqb) / p.nwg, nq = ((wg + 1) * p.nqb) / p.nwg - qb0;      // nq <= 8
    const int nmine = (wave < nq) + (wave + 4 < nq);                                     // 0, 1 or 2 (wave-uniform)

    // ---- staging (4 waves: each thread moves 2 K chunks + 2 V^T chunks per tile)
    const int srow = tid >> 3;
    const int lslot = (tid & 7) ^ ((srow >> 1) & 7);
    const int m_begin = tok0 & ~63;
    const int ntile = (((tok1 + 63) & ~63) - m_begin) >> 6;
    const bf16_t* kcur = p.k + h * 64 + lslot * 8 + (size_t)(m_begin + srow) * p.ldk;
    const bf16_t* vcur = p.vt + (size_t)(h * 64 + srow) * p.ldvt + lslot * 8 + m_begin;
    const size_t kstep = (size_t)KT * p.ldk, khalf = (size_t)32 * p.ldk, vhalf = (size_t)32 * p.ldvt;
    int mt_k = m_begin;
    char* const kring = smem;
    char* const vring = smem + NSLOT * TILE_B;
    auto stage_k = [&](int slot) {                         // next K tile -> kring[slot]
        char* sk = kring + slot * TILE_B + wave * 1024;
        if (mt_k + KT <= p.Mk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(kcur + j * khalf, sk + j * 4096);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int mk = mt_k + j * 32 + srow;       // rows past the last token are masked: re-read the last row
                const bf16_t* kr = mk < p.Mk ? kcur + j * khalf : kcur - (size_t)(mt_k + srow - (p.Mk - 1)) * p.ldk;
                glds16(kr, sk + j * 4096);
            }
        }
        kcur += kstep; mt_k += KT;
    };
    auto stage_v = [&](int slot) {                         // next V^T tile -> vring[slot]
        char* sv = vring + slot * TILE_B + wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(vcur + j * vhalf, sv + j * 4096);
        vcur += KT;
    };

    const int rsw = (lq >> 1) & 7;
    const int rbase = lq * 128;

    // prologue: K(0), V(0), K(1) in flight
    stage_k(0); stage_v(0);
    if (ntile > 1) stage_k(1);

    QBlk A, Bq;
    char* const qtiles = smem + 2 * NSLOT * TILE_B;        // 4 waves x 2 query blocks x 4 KB
    if (nmine >= 1) qblk_init(A, p, b, h, qb0 + wave, lq, hi, qtiles + (wave * 2 + 0) * 4096);
    if (nmine >= 2) qblk_init(Bq, p, b, h, qb0 + 4 + wave, lq, hi, qtiles + (wave * 2 + 1) * 4096);
    __syncthreads();

    // mask parameters of tile t for a query block: keys [klo, klo + krange) of the tile's 64 are valid (this lane's rows start at 4 * hi)
    auto mask_of = [&](int t, const QBlk& X, int& klo, int& krange) -> bool {
        const int mt = m_begin + t * KT;
        int lo = tok0 - mt, hb = tok1 - mt;
        if (p.causal) hb = min(hb, tok0 + X.qloc + 1 - mt);
        lo = max(lo, 0); hb = min(hb, 64);
        klo = lo - 4 * hi; krange = max(hb - lo, 0);
        return (mt < tok0) || (mt + KT > tok1) || p.causal;                               // wave-uniform
    };

    if (nmine == 0) {
        for (int t = 0; t < ntile; ++t) {
            if (t + 2 < ntile) stage_k((t + 2) % NSLOT);
            if (t + 1 < ntile) stage_v((t + 1) % NSLOT);
            __syncthreads();
        }
        return;
    }
    int klo = 0, krange = 0;
    // One key tile of the two-block stream.  FIRST: no P_B yet; LAST: no next tile for A.  (The first / steady / last tiles are three
    // straight-line copies: runtime `t == 0` selects inside one loop body made hipcc carry both variants' values through the loop.)
    auto tile2 = [&](int t, auto first, auto last) {
        constexpr bool FIRST = decltype(first)::value, LAST = decltype(last)::value;
        if (t + 2 < ntile) stage_k((t + 2) % NSLOT);
        if (t + 1 < ntile) stage_v((t + 1) % NSLOT);
        const char* sk0 = kring + (t % NSLOT) * TILE_B;
        const char* sk1 = kring + ((t + 1) % NSLOT) * TILE_B;
        const char* sv0 = vring + ((t + NSLOT - 1) % NSLOT) * TILE_B;
        const char* sv1 = vring + (t % NSLOT) * TILE_B;
        // phase 1: softmax A(t) beside O_B += V(t-1) P_B(t-1) and S_B(t)
        if (mask_of(t, A, klo, krange)) mask_scores(A, klo, krange);
        ab_phase<true, !FIRST, true>(Bq, A, sk0, sv0, p.sc, rbase, rsw, hi);
        // phase 2: softmax B(t) beside O_A += V(t) P_A(t) and S_A(t+1)
        if (mask_of(t, Bq, klo, krange)) mask_scores(Bq, klo, krange);
        ab_phase<!LAST, true, true>(A, Bq, sk1, sv1, p.sc, rbase, rsw, hi);
        __syncthreads();
    };
    auto tile1 = [&](int t, auto last) {                   // the same for a wave with one query block
        constexpr bool LAST = decltype(last)::value;
        if (t + 2 < ntile) stage_k((t + 2) % NSLOT);
        if (t + 1 < ntile) stage_v((t + 1) % NSLOT);
        const char* sk1 = kring + ((t + 1) % NSLOT) * TILE_B;
        const char* sv1 = vring + (t % NSLOT) * TILE_B;
        if (mask_of(t, A, klo, krange)) mask_scores(A, klo, krange);
        ab_phase<false, false, true>(A, A, sk1, sv1, p.sc, rbase, rsw, hi);
        ab_phase<!LAST, true, false>(A, A, sk1, sv1, p.sc, rbase, rsw, hi);
        __syncthreads();
    };
    using T_ = std::true_type; using F_ = std::false_type;
    // S_A(0)
    ab_phase<true, false, false>(A, A, kring, vring, p.sc, rbase, rsw, hi);
    if (nmine == 2) {
        if (ntile == 1) tile2(0, T_{}, T_{});
        else {
            tile2(0, T_{}, F_{});
            for (int t = 1; t + 1 < ntile; ++t) tile2(t, F_{}, F_{});
            tile2(ntile - 1, F_{}, T_{});
        }
        // O_B += V(last) P_B(last)
        ab_phase<false, true, false>(Bq, Bq, kring, vring + ((ntile - 1) % NSLOT) * TILE_B, p.sc, rbase, rsw, hi);
        qblk_store(A, p, b, h, hi);
        qblk_store(Bq, p, b, h, hi);
    } else {
        for (int t = 0; t + 1 < ntile; ++t) tile1(t, F_{});
        tile1(ntile - 1, T_{});
        qblk_store(A, p, b, h, hi);
    }
}

}  // namespace

int visrep_attention_ab_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* out, int ldo,
                               int B, int Tq, int Tk, int H, int kv_shared, int causal, float scale, hipStream_t st) {
    AttnArgs2 a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out;
    const long Mk = kv_shared ? Tk : (long)B * Tk;
    a.B = B; a.Tq = Tq; a.Tk = Tk; a.H = H; a.Mk = (int)Mk; a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo; a.kv_shared = kv_shared; a.causal = causal;
    a.sc = scale * 1.4426950408889634f;
    a.nqb = (Tq + 31) / 32;
    a.nwg = (a.nqb + 7) / 8;
    const dim3 grid(a.nwg * H * B), block(256);
    const size_t lds = (size_t)2 * NSLOT * TILE_B + 8 * 4096;   // K ring + V^T ring + 8 Q tiles = 80 KB: two workgroups per CU
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_ab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(attn_fwd_ab, grid, block, lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "attention: launch failed");
}
