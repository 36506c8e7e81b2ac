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
//   * an image-head's query blocks go to workgroups of eight (two per wave) and the remainder to workgroups whose waves take one
//     block each (577 rows = 19 blocks -> 8 + 8 + 3): such a wave runs the same phases without the partner stream, a wave with no
//     block only stages.
// Everything else (swapped S^T, in-register P, perm16 V^T layout, global 64-key tiles with masked edges, XCD remap, the
// lane-swap-widened stores) is attn_fwd<1>'s and is described there.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "visrep_internal.h"

// Deferred running maximum: the old maximum is kept while no row's maximum grew by more than T (in exp2 units), so P <= 2^T and
// the 32 rescale multiplies per block and tile leave the stream (the rare rescale runs in a cold branch).  bf16 P keeps its relative
// precision at any magnitude, the row sum is fp32: the result moves by the rounding of P only (2.9e-3 rel-L2 against attn_fwd<1>).
#ifndef VISREP_ATTN_THR
#define VISREP_ATTN_THR 8
#endif
static_assert(VISREP_ATTN_THR > 0, "attn_fwd_ab needs the deferred maximum");
// how many fragment reads run ahead of the MFMA that consumes them (4 registers each)
#ifndef VISREP_ATTN_AB_DEPTH
#define VISREP_ATTN_AB_DEPTH 5
#endif
// timing-only: 1 = drop the sched_barrier pins (let hipcc order the phase)
#ifndef VISREP_ATTN_AB_NOPIN
#define VISREP_ATTN_AB_NOPIN 0
#endif

// Timing-only ablations (tools/attn_ablate.py; results are WRONG for mask != 0): 1 no v_exp, 2 no MFMAs, 4 no fragment reads / waits,
// 8 no barrier and no LDS-DMA in the tile loop, 16 no score pairs (fma / exp / add / cvt), 32 no running maximum / decision / rescale,
// 64 no maximum chains (decision on one element), 128 maximum chains but no decision (no lane swap, no branch, no rescale)
#ifndef VISREP_ATTN_AB_ABLATE
#define VISREP_ATTN_AB_ABLATE 0
#endif
#define ABL(bit) ((VISREP_ATTN_AB_ABLATE & (bit)) != 0)
// A/B builds of single choices: 1 one counted wait per MFMA (instead of per MFMA pair), 2 plain fp32 fma / add per score (instead of the
// packed forms)
#ifndef VISREP_ATTN_AB_ALT
#define VISREP_ATTN_AB_ALT 0
#endif
#define ALT(bit) ((VISREP_ATTN_AB_ALT & (bit)) != 0)

namespace {

constexpr int KT = 64;
constexpr int TILE_B = KT * 64 * 2;        // 8 KB
constexpr int NSLOT = 3;                   // ring depth of the K ring and of the V^T ring

struct AttnArgs2 {
    const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* out;
    int B, Tq, Tk, H, Mk, ldq, ldk, ldvt, ldo, kv_shared, causal;
    int nqb, nwg, nfull;                   // 32-row query blocks per image-head, workgroups per image-head, of which with 8 blocks
    float sc;
};

struct QBlk {
    f32x16 o[2];          // O^T accumulators (d tiles of 32)
    f32x16 s[2];          // raw scores of the tile in flight (two 32-key halves)
    uint32_t pb[2][8];    // bf16-packed P of the last softmax
    float m, l;           // running maximum (raw score units) and running sum of this lane's keys
    int qloc;             // this lane's query row inside the image
    unsigned qa[4];       // LDS byte addresses of this lane's four Q fragments (this block's Q tile: 32 rows x 128 B, swizzled like a K tile)
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

// LDS fragment read / counted wait as asm: hipcc then knows nothing of the outstanding reads, so the only waits in a phase are the
// ones written here - lgkmcnt(n) with n = the number of YOUNGER reads still allowed in flight - and the reads run VISREP_ATTN_AB_DEPTH
// fragments ahead of the MFMA that consumes them (compiler-placed waits were lgkmcnt(0 / 1): every MFMA stood behind the LDS
// latency of its own operand).  The wait names its fragment as an in-out operand, which orders it between the read and the MFMA.
VR_DEV void lds_read_asm(bf16x8& f, unsigned addr, int half) {
    if (half) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f) : "v"(addr));
    else asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(addr));
}
// The wait names every fragment the MFMAs behind it consume as in-out operands: that is what orders read < wait < MFMA for hipcc.
// (Never name one fragment twice: the second in-out operand would be a register COPY taken before the wait - stale data.)
#define VR_WAIT_CASES(OPS) \
    switch (n) { \
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : OPS); break; case 1: asm volatile("s_waitcnt lgkmcnt(1)" : OPS); break; \
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : OPS); break; case 3: asm volatile("s_waitcnt lgkmcnt(3)" : OPS); break; \
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : OPS); break; case 5: asm volatile("s_waitcnt lgkmcnt(5)" : OPS); break; \
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" : OPS); break; case 7: asm volatile("s_waitcnt lgkmcnt(7)" : OPS); break; \
        case 8: asm volatile("s_waitcnt lgkmcnt(8)" : OPS); break; case 9: asm volatile("s_waitcnt lgkmcnt(9)" : OPS); break; \
        case 10: asm volatile("s_waitcnt lgkmcnt(10)" : OPS); break; case 11: asm volatile("s_waitcnt lgkmcnt(11)" : OPS); break; \
        default: asm volatile("s_waitcnt lgkmcnt(12)" : OPS); break; \
    }
#define VR_COMMA ,
VR_DEV void lds_wait_asm(bf16x8& f0, int n) { VR_WAIT_CASES("+v"(f0)) }
VR_DEV void lds_wait_asm(bf16x8& f0, bf16x8& f1, int n) { VR_WAIT_CASES("+v"(f0) VR_COMMA "+v"(f1)) }
VR_DEV void lds_wait_asm(bf16x8& f0, bf16x8& f1, bf16x8& f2, int n) { VR_WAIT_CASES("+v"(f0) VR_COMMA "+v"(f1) VR_COMMA "+v"(f2)) }
#undef VR_WAIT_CASES
#undef VR_COMMA

// One phase.  M side (matrix): DO_PV: M.o += V^T-tile(sv) P_M, then DO_QK: M.s = K-tile(sk) Q_M^T (Q fragments from M's LDS tile).
// V side (softmax, DO_SM): V.s -> V.pb, V.m, V.l, V.o rescaled.  sk / sv: LDS byte offsets of the tiles; off[kk]: this lane's
// fragment offset inside any 64-row tile (row lq, swizzled 16-byte slot 2 kk + hi).
template <bool DO_QK, bool DO_PV, bool DO_SM>
VR_DEV void ab_phase(QBlk& M, QBlk& V, unsigned sk, unsigned sv, float sc, const unsigned (&off)[4]) {
    constexpr int NM = (DO_QK ? 8 : 0) + (DO_PV ? 8 : 0);
    constexpr int NS = NM ? NM : 1;                       // slots
    constexpr int QK0 = DO_PV ? 8 : 0;                    // first Q.K^T op = number of V^T fragment reads
    constexpr int NR = QK0 + (DO_QK ? 12 : 0);            // fragment reads of the phase, in consumption order:
    constexpr int DEPTH = VISREP_ATTN_AB_DEPTH;           //   V^T(c, dt) x 8, then per kk: Q(kk), K(kk, 0), K(kk, 1)

    // ---- M side.  Matrix op order: the eight P.V MFMAs first (P dies as they go), then the eight Q.K^T ones (S is born late).
    bf16x8 rd[NR ? NR : 1];
    unsigned va[4], ka[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { va[kk] = off[kk] + sv; ka[kk] = off[kk] + sk; }
    auto issue_read = [&](int r) {
        if (r < QK0) lds_read_asm(rd[r], va[r >> 1], r & 1);
        else {
            const int kk = (r - QK0) / 3, w = (r - QK0) % 3;
            if (w == 0) lds_read_asm(rd[r], M.qa[kk], 0);
            else lds_read_asm(rd[r], ka[kk], w - 1);
        }
    };
    auto last_read_of = [&](int j) -> int {               // index of the youngest read matrix op j consumes
        if (!DO_QK || j < QK0) return j;
        const int jj = j - QK0;
        return QK0 + 3 * (jj >> 1) + 1 + (jj & 1);
    };
    int issued = 0;
    if (ABL(4)) {
#pragma unroll
        for (int r = 0; r < (NR ? NR : 1); ++r) rd[r] = bf16x8{(short)(sk + r), 1, 2, 3, (short)sv, 5, 6, 7};
    }
    auto feed = [&](int j) {                              // even j: reads up to DEPTH past the operands of ops j and j + 1, then ONE wait for both
        if (ABL(4) || (!ALT(1) && (j & 1))) return;
        const int need = last_read_of(ALT(1) ? j : j + 1) + 1;
        const int target = need + DEPTH < NR ? need + DEPTH : NR;
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (r >= issued && r < target) issue_read(r);
        issued = target > issued ? target : issued;
        // ops j, j + 1 consume reads need - 2, need - 1 (P.V) or need - 3 .. need - 1 (Q.K^T: Q fragment, two K fragments)
        if (ALT(1)) lds_wait_asm(rd[need - 1], issued - need);    // (a Q fragment was waited for with the first K fragment behind it)
        else if (DO_QK && j >= QK0) lds_wait_asm(rd[need - 1], rd[need - 2], rd[need - 3], issued - need);
        else lds_wait_asm(rd[need - 1], rd[need - 2], issued - need);
    };
    auto mat_op = [&](int j) {
        const bool qk = DO_QK && j >= QK0;
        const int jj = qk ? j - QK0 : j;
        if (ABL(2)) {
            if (qk) { if ((jj >> 1) == 0) M.s[jj & 1] = f32x16{}; M.s[jj & 1][jj] += __builtin_bit_cast(float, (int)rd[ABL(4) ? 0 : QK0 + 3 * (jj >> 1) + 1 + (jj & 1)][0] + (int)rd[ABL(4) ? 0 : QK0 + 3 * (jj >> 1)][1]); }
            else M.o[jj & 1][jj] += __builtin_bit_cast(float, (int)rd[ABL(4) ? 0 : j][0] + (int)M.pb[jj >> 2][jj & 7]);
            return;
        }
        if (qk) {
            const int kk = jj >> 1, kt2 = jj & 1;
            const f32x16 acc = kk == 0 ? f32x16{} : M.s[kt2];
            M.s[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd[QK0 + 3 * kk + 1 + kt2], rd[QK0 + 3 * kk], acc, 0, 0, 0);
        } else {
            const int c = jj >> 1, dt = jj & 1;
            u32x4 w = {M.pb[c >> 1][4 * (c & 1) + 0], M.pb[c >> 1][4 * (c & 1) + 1], M.pb[c >> 1][4 * (c & 1) + 2], M.pb[c >> 1][4 * (c & 1) + 3]};
            M.o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd[j], *reinterpret_cast<bf16x8*>(&w), M.o[dt], 0, 0, 0);
        }
    };

    // ---- V side: the softmax of V.s cut into slot-sized pieces
    float mloc0, mloc1, nmsc = 0.f, alpha = 1.f;
    f32x2 psa = {0.f, 0.f}, psb = {0.f, 0.f};
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
        if (__any((mloc - V.m) * sc > (float)VISREP_ATTN_THR)) {      // wave-uniform; V.m = -inf on the first tile -> taken
            const float m_new = fmaxf(V.m, mloc);
            alpha = __builtin_amdgcn_exp2f((V.m - m_new) * sc);
            V.m = m_new;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) V.o[dt][r] *= alpha;
        }
        nmsc = -V.m * sc;
    };
    // Score elements in groups of eight, as asm (plain C++ lets hipcc sink all thirty-two row-sum adds to the end of the phase, which
    // keeps every exponential alive - 32 registers - and un-does the interleave).  A wave issues one instruction per ~6 cycles
    // whatever its kind (tools/probes/valu_probe.hip), so the scale-and-shift and the row sums use the packed fp32 forms: 20
    // instructions per eight scores instead of 28 (same VALU time, fewer issue slots).
    const f32x2 sc2 = {sc, sc};
    auto sm_octet = [&](int g) {                          // score elements 8g .. 8g + 7 of 32
        const int kt2 = g >> 1, r = (g & 1) * 8;
        if (ABL(16)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) V.pb[kt2][(r >> 1) + j] = __builtin_bit_cast(unsigned, V.s[kt2][r + 2 * j]);
            return;
        }
        if (ALT(2)) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                float a0, a1;
                asm volatile("v_fma_f32 %0, %5, %7, %8\n\t"
                             "v_fma_f32 %1, %6, %7, %8\n\t"
                             "v_exp_f32 %0, %0\n\t"
                             "v_exp_f32 %1, %1\n\t"
                             "v_add_f32 %2, %2, %0\n\t"
                             "v_add_f32 %3, %3, %1\n\t"
                             "v_cvt_pk_bf16_f32 %4, %0, %1"
                             : "=&v"(a0), "=&v"(a1), "+v"(psa[0]), "+v"(psa[1]), "=v"(V.pb[kt2][(r + j) >> 1])
                             : "v"(V.s[kt2][r + j]), "v"(V.s[kt2][r + j + 1]), "s"(sc), "v"(nmsc));
            }
            return;
        }
        const f32x2 s01 = {V.s[kt2][r], V.s[kt2][r + 1]}, s23 = {V.s[kt2][r + 2], V.s[kt2][r + 3]};
        const f32x2 s45 = {V.s[kt2][r + 4], V.s[kt2][r + 5]}, s67 = {V.s[kt2][r + 6], V.s[kt2][r + 7]};
        const f32x2 nm2 = {nmsc, nmsc};
        // v[248:255] are this statement's temporaries, named outright (and listed as clobbers) because an asm operand cannot be
        // addressed by halves: the packed instructions see them as four pairs, v_exp / v_cvt_pk as eight scalars.  Every producer is
        // at least three instructions ahead of its first consumer (a packed result's upper half and a transcendental's result are
        // not forwarded to the very next instruction).
        asm volatile("v_pk_fma_f32 v[248:249], %6, %10, %11\n\t"
                     "v_pk_fma_f32 v[250:251], %7, %10, %11\n\t"
                     "v_pk_fma_f32 v[252:253], %8, %10, %11\n\t"
                     "v_pk_fma_f32 v[254:255], %9, %10, %11\n\t"
#if !(VISREP_ATTN_AB_ABLATE & 1)
                     "v_exp_f32 v248, v248\n\t"
                     "v_exp_f32 v249, v249\n\t"
                     "v_exp_f32 v250, v250\n\t"
                     "v_exp_f32 v251, v251\n\t"
                     "v_exp_f32 v252, v252\n\t"
                     "v_exp_f32 v253, v253\n\t"
                     "v_exp_f32 v254, v254\n\t"
                     "v_exp_f32 v255, v255\n\t"
#endif
                     "v_pk_add_f32 %0, %0, v[248:249]\n\t"
                     "v_pk_add_f32 %1, %1, v[250:251]\n\t"
                     "v_pk_add_f32 %0, %0, v[252:253]\n\t"
                     "v_pk_add_f32 %1, %1, v[254:255]\n\t"
                     "v_cvt_pk_bf16_f32 %2, v248, v249\n\t"
                     "v_cvt_pk_bf16_f32 %3, v250, v251\n\t"
                     "v_cvt_pk_bf16_f32 %4, v252, v253\n\t"
                     "v_cvt_pk_bf16_f32 %5, v254, v255"
                     : "+v"(psa), "+v"(psb), "=v"(V.pb[kt2][(r >> 1) + 0]), "=v"(V.pb[kt2][(r >> 1) + 1]), "=v"(V.pb[kt2][(r >> 1) + 2]), "=v"(V.pb[kt2][(r >> 1) + 3])
                     : "v"(s01), "v"(s23), "v"(s45), "v"(s67), "v"(sc2), "v"(nm2)
                     : "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
    };
    // Sixteen-slot plan: slot 0 carries no softmax work (the Q.K^T that produced V.s ended the previous phase: its result latency
    // passes under this phase's first MFMA), two slots of running maximum, one decision slot, then a score octet every third slot.
    auto sm_slot16 = [&](int i) {
        if (ABL(32)) { if (i == 3) { nmsc = -V.s[0][0]; alpha = 1.f; } }
        else if (i == 1) { if (ABL(64)) mloc0 = V.s[0][0]; else sm_max_a(); }
        else if (i == 2) { if (!ABL(64)) sm_max_b(); }
        else if (i == 3) { if (ABL(128)) { nmsc = -mloc0 * sc; alpha = 1.f; } else sm_decide(); }
        else if (i > 3) {
            const int k = i - 4;                          // 0 .. 11
            if (k % 3 == 0) sm_octet(k / 3);
        }
    };

#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (NM) {
            feed(i);
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
    if (DO_SM) V.l = __builtin_fmaf(V.l, alpha, (psa[0] + psa[1]) + (psb[0] + psb[1]));
}

// Q rows of one query block: global -> registers -> this block's swizzled LDS tile (read back as MFMA B fragments every tile).
VR_DEV void qblk_init(QBlk& X, const AttnArgs2& p, int b, int h, int qb, int lq, int hi, char* smem, unsigned sq) {
    X.qloc = qb * 32 + lq;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) X.qa[kk] = sq + lq * 128 + (((2 * kk + hi) ^ ((lq >> 1) & 7)) << 4);
    const size_t qrow = (size_t)b * p.Tq + (X.qloc < p.Tq ? X.qloc : p.Tq - 1);
    const bf16_t* qg = p.q + qrow * p.ldq + h * 64 + hi * 8;
    const int rsw = (lq >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        *reinterpret_cast<bf16x8*>(smem + sq + lq * 128 + (((2 * kk + hi) ^ rsw) << 4)) = *reinterpret_cast<const bf16x8*>(qg + kk * 16);
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

    // Query blocks of this workgroup.  The waves of a workgroup meet at a barrier every key tile, so they must carry EQUAL work: the
    // first nfull workgroups of an image-head take eight blocks (two per wave), the remainder goes to one or two workgroups whose
    // waves take ONE block each (577 rows = 19 blocks -> 8 + 8 + 3; with the even split 7 + 6 + 6 the one-block waves of every
    // workgroup spent half of their time parked at the barrier: 21 % of all wave cycles, profiles/round3_attention.md).
    const int qb0 = wg < p.nfull ? 8 * wg : 8 * p.nfull + 4 * (wg - p.nfull);
    const int nq = wg < p.nfull ? 8 : min(4, p.nqb - qb0);
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

    unsigned off[4];                                       // fragment offset inside a 64-row tile: row lq, swizzled slot 2 kk + hi
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) off[kk] = lq * 128 + (((2 * kk + hi) ^ ((lq >> 1) & 7)) << 4);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);   // LDS byte address of the dynamic segment (uniform)
    const unsigned kring_o = lds0, vring_o = lds0 + NSLOT * TILE_B, qtiles_o = lds0 + 2 * NSLOT * TILE_B;

    // prologue: K(0), V(0), K(1) in flight
    stage_k(0); stage_v(0);
    if (ntile > 1) stage_k(1);

    QBlk A, Bq;
    // Q tiles: 4 waves x 2 query blocks x 4 KB behind the rings (LDS addresses of a dynamic-only kernel start at 0: lds0 == 0)
    if (nmine >= 1) qblk_init(A, p, b, h, qb0 + wave, lq, hi, smem - lds0, qtiles_o + (wave * 2 + 0) * 4096);
    if (nmine >= 2) qblk_init(Bq, p, b, h, qb0 + 4 + wave, lq, hi, smem - lds0, qtiles_o + (wave * 2 + 1) * 4096);
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
            if (!ABL(8) && t + 2 < ntile) stage_k((t + 2) % NSLOT);
            if (!ABL(8) && t + 1 < ntile) stage_v((t + 1) % NSLOT);
            if (!ABL(8)) __syncthreads();
        }
        return;
    }
    int klo = 0, krange = 0;
    // One key tile of the two-block stream.  FIRST: no P_B yet; LAST: no next tile for A.  (The first / steady / last tiles are three
    // straight-line copies: runtime `t == 0` selects inside one loop body made hipcc carry both variants' values through the loop.)
    auto tile2 = [&](int t, auto first, auto last) {
        constexpr bool FIRST = decltype(first)::value, LAST = decltype(last)::value;
        if (!ABL(8) && t + 2 < ntile) stage_k((t + 2) % NSLOT);
        if (!ABL(8) && t + 1 < ntile) stage_v((t + 1) % NSLOT);
        const unsigned sk0 = kring_o + (t % NSLOT) * TILE_B, sk1 = kring_o + ((t + 1) % NSLOT) * TILE_B;
        const unsigned sv0 = vring_o + ((t + NSLOT - 1) % NSLOT) * TILE_B, sv1 = vring_o + (t % NSLOT) * TILE_B;
        // phase 1: softmax A(t) beside O_B += V(t-1) P_B(t-1) and S_B(t)
        if (mask_of(t, A, klo, krange)) mask_scores(A, klo, krange);
        ab_phase<true, !FIRST, true>(Bq, A, sk0, sv0, p.sc, off);
        // phase 2: softmax B(t) beside O_A += V(t) P_A(t) and S_A(t+1)
        if (mask_of(t, Bq, klo, krange)) mask_scores(Bq, klo, krange);
        ab_phase<!LAST, true, true>(A, Bq, sk1, sv1, p.sc, off);
        if (!ABL(8)) __syncthreads();
    };
    auto tile1 = [&](int t, auto last) {                   // the same for a wave with one query block
        constexpr bool LAST = decltype(last)::value;
        if (!ABL(8) && t + 2 < ntile) stage_k((t + 2) % NSLOT);
        if (!ABL(8) && t + 1 < ntile) stage_v((t + 1) % NSLOT);
        const unsigned sk1 = kring_o + ((t + 1) % NSLOT) * TILE_B, sv1 = vring_o + (t % NSLOT) * TILE_B;
        if (mask_of(t, A, klo, krange)) mask_scores(A, klo, krange);
        ab_phase<false, false, true>(A, A, sk1, sv1, p.sc, off);
        ab_phase<!LAST, true, false>(A, A, sk1, sv1, p.sc, off);
        if (!ABL(8)) __syncthreads();
    };
    using T_ = std::true_type; using F_ = std::false_type;
    // S_A(0)
    ab_phase<true, false, false>(A, A, kring_o, vring_o, p.sc, off);
    if (nmine == 2) {
        if (ntile == 1) tile2(0, T_{}, T_{});
        else {
            tile2(0, T_{}, F_{});
            for (int t = 1; t + 1 < ntile; ++t) tile2(t, F_{}, F_{});
            tile2(ntile - 1, F_{}, T_{});
        }
        // O_B += V(last) P_B(last)
        ab_phase<false, true, false>(Bq, Bq, kring_o, vring_o + ((ntile - 1) % NSLOT) * TILE_B, p.sc, off);
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
    a.nfull = a.nqb / 8;
    a.nwg = a.nfull + ((a.nqb & 7) ? ((a.nqb & 7) > 4 ? 2 : 1) : 0);
    const dim3 grid(a.nwg * H * B), block(256);
    const size_t lds = (size_t)2 * NSLOT * TILE_B + 8 * 4096;   // K ring + V^T ring + 8 Q tiles = 80 KB: two workgroups per CU
    static VisrepLdsOptIn opt;
    visrep_lds_opt_in(opt, (const void*)attn_fwd_ab, (int)lds);
    static int told = 0;
    if (!told && getenv("VISREP_DEBUG")) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)attn_fwd_ab, 256, lds);
        fprintf(stderr, "[visrep] attn_fwd_ab: %zu B LDS per workgroup, %d workgroups per CU, grid %u\n", lds, nb, grid.x);
        told = 1;
    }
    hipLaunchKernelGGL(attn_fwd_ab, grid, block, lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "attention: launch failed");
}
