// Fused multi-head attention forward on gfx950 (no mask, full softmax): ViT self-attention (head width 64) and the
// UNet self- / cross-attention of the diffusion towers (head width D = 64*ND, narrower heads zero-padded by the weight
// packer; key/value sequence of its own length Tk, optionally SHARED by all batch items - the prompt).
//
//   O[b, q, h, :] = softmax_k( Q[b,q,h,:] . K[b,k,h,:] * scale ) V[b,k,h,:]      (causal: keys k <= q only - CLIP text encoder)
//
// Layouts (produced by the GEMM epilogues, gemm_bf16.hip):
//   q   : [B*Tq, ldq] bf16 row-major, head h at columns h*D;  k : [B*Tk (or Tk), ldk] likewise
//         (ViT: one [M, 2d] buffer, Q in columns [0,d), K in columns [d,2d))
//   vt  : [H*D, ldvt] bf16, row n = h*D+dd, column perm16(m) over the GLOBAL key index m = b*Tk + t.
//         perm16 swaps the two middle 4-token groups of every aligned 16-token block (0,2,1,3), so that the eight
//         keys one lane needs for a 32x32x16 MFMA k-slice are 16 contiguous bytes (see below).
//   out : [M, d] bf16 row-major (heads concatenated), feeds the out-projection GEMM.
//
// Design (flash style, online softmax, fp32 statistics):
//   * block = 4 waves, 128 query rows of one (b, h); each wave owns 32 query rows
//   * S^T = K Q^T is computed "swapped" with v_mfma_f32_32x32x16_bf16 (A = K tile from LDS, B = Q fragments held in
//     registers for the whole kernel): lane l then holds, for ITS query q = l&31, sixteen keys per 32-key tile
//     (row(r) = (r&3) + 8*(r>>2) + 4*(l>>5)).  Row max / sum need one cross-lane exchange (l ^ 32) only.
//   * O^T += V^T P^T reuses the S^T registers directly as the B operand (bf16-packed in place): the MFMA k index is
//     only a summation index, so any permutation applied to BOTH operands is legal; the lane's keys are
//     {16c+4hi+0..3, 16c+8+4hi+0..3}, which perm16 makes contiguous in the V^T rows -> one ds_read_b128 per operand.
//   * K / V^T tiles (64 keys) stream through a double-buffered, XOR-swizzled LDS ring filled by global_load_lds;
//     key tiles are aligned to GLOBAL 64-token blocks, so no per-image padding exists anywhere: the first and last
//     tile of an image are masked against [b*T, (b+1)*T).
//   * blockIdx is XCD-remapped so the q-tiles of one (b,h) share an XCD L2 (K/V fetched from HBM once).
#include <cstdlib>

#include "common.h"
#include "visrep_internal.h"

// Timing-only ablation builds (-DVISREP_ATTN_ABLATE=mask, tools/attn_ablate.py): results are WRONG for mask != 0.
//   1: no v_exp (p = the FMA result)   2: no P.V MFMAs   4: no Q.K^T MFMAs   8: no LDS fragment reads (registers reused)
//   16: no per-tile barrier / next-tile DMA (tile 0 is re-used)
#ifndef VISREP_ATTN_ABLATE
#define VISREP_ATTN_ABLATE 0
#endif
#ifndef VISREP_ATTN_V1_THR
#define VISREP_ATTN_V1_THR 8             // 0: update the running maximum on every tile
#endif

namespace {

constexpr int KT = 64;                    // keys per tile
constexpr int TILE_B = KT * 64 * 2;       // 8 KB: one [64 keys x 64 d] K sub-tile or [64 d x 64 keys] V^T sub-tile

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* out;
    int B, Tq, Tk, H, Mk, ldq, ldk, ldvt, ldo, kv_shared, causal;
    float sc;                              // softmax scale * log2(e)
};

// PS ("pre-scaled", round 4): the caller folded scale * log2(e) into Q (the ViT engine folds it into the Q rows of the Q|K projection
// weights), so the raw MFMA result is already the exponent of 2 - and the running reference maximum is subtracted INSIDE the matrix pipe:
// the Q.K^T accumulators start from a register tile holding -m_ref instead of zeros, p = exp2(acc) with no per-element scale-and-shift FMA
// (32 of the ~250 VALU instructions per key tile, profiles/round4_attention.md).  The reference is moved (scores shifted, O and the running
// sum rescaled, the init tile rewritten) only on an image's first key tile and when some row's maximum grew by more than 2^THR.
template <int ND, bool PS = false>         // head width D = 64 * ND
__global__ __launch_bounds__(256, ND == 1 ? 2 : 1) void attn_fwd(const AttnArgs p) {
    constexpr int D = 64 * ND, KV_B = ND * TILE_B, STAGE_B = 2 * KV_B;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const int nqt = (p.Tq + 127) >> 7;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int h = id % p.H;
    const int b = id / p.H;
    const int tok0 = p.kv_shared ? 0 : b * p.Tk, tok1 = tok0 + p.Tk;   // this item's global key range

    // ---- Q fragments (B operand: col q = lane&31, k = d index 16*kk + 8*hi .. +8), kept in registers
    const int qloc = qt * 128 + wave * 32 + lq;
    const size_t qrow = (size_t)b * p.Tq + (qloc < p.Tq ? qloc : p.Tq - 1);
    bf16x8 qf[4 * ND];
#pragma unroll
    for (int kk = 0; kk < 4 * ND; ++kk)
        qf[kk] = *reinterpret_cast<const bf16x8*>(p.q + qrow * p.ldq + h * D + kk * 16 + hi * 8);

    // ---- staging: 64 rows x 8 slots per tile = 2 chunks per thread per tile
    const int srow = tid >> 3;
    const int lslot = (tid & 7) ^ ((srow >> 1) & 7);
    const int m_begin = tok0 & ~63;
    const int ntile = (((tok1 + 63) & ~63) - m_begin) >> 6;
    // Running source pointers of this thread's chunks (tile `it` is staged right after tile it - 1, so they only ever advance by one
    // tile: 64 key rows of K, 64 key columns of V^T); only a tile that reaches past the last key row of the buffer re-derives clamped rows.
    const bf16_t* kcur = p.k + h * D + lslot * 8 + (size_t)(m_begin + srow) * p.ldk;
    const bf16_t* vcur = p.vt + (size_t)(h * D + srow) * p.ldvt + lslot * 8 + m_begin;
    const size_t kstep = (size_t)KT * p.ldk, khalf = (size_t)32 * p.ldk, vhalf = (size_t)32 * p.ldvt;
    int mt_stage = m_begin;
    auto stage = [&](int buf) {
        char* sk = smem + buf * STAGE_B + wave * 1024;
        char* sv = sk + KV_B;
        if (mt_stage + KT <= p.Mk) {                        // uniform
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nd = 0; nd < ND; ++nd) glds16(kcur + j * khalf + nd * 64, sk + nd * TILE_B + j * 4096);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int mk = mt_stage + j * 32 + srow;    // rows past the last token are masked below: re-read the last row
                const bf16_t* kr = mk < p.Mk ? kcur + j * khalf : kcur - (size_t)(mt_stage + srow - (p.Mk - 1)) * p.ldk;
#pragma unroll
                for (int nd = 0; nd < ND; ++nd) glds16(kr + nd * 64, sk + nd * TILE_B + j * 4096);
            }
        }
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(vcur + nd * 2 * vhalf + j * vhalf, sv + nd * TILE_B + j * 4096);
        kcur += kstep; vcur += KT; mt_stage += KT;
    };

    // fragment read offsets inside a tile: row = 32*blk + lq, logical slot s -> physical s ^ ((lq>>1)&7)
    const int rsw = (lq >> 1) & 7;
    const int rbase = lq * 128;

    f32x16 o[2 * ND];
#pragma unroll
    for (int dt = 0; dt < 2 * ND; ++dt) o[dt] = f32x16{};
    float m_run = PS ? 0.f : -INFINITY, l_run = 0.f;       // PS: m_run = the reference already subtracted in the accumulators
    f32x16 minit = f32x16{};                               // PS: -m_run in every element (the C operand of the first Q.K^T MFMA of a chain)

    stage(0);
    __syncthreads();
    if (qt * 128 + wave * 32 >= p.Tq) {
        // this wave's 32 query rows are all past the sequence (the last query tile of T = 577 has 65 rows: wave 3 is empty):
        // it only keeps staging its quarter of the K / V^T tiles and meeting the barriers, and leaves its SIMD to other blocks
        for (int it = 0; it < ntile; ++it) {
            if (!(VISREP_ATTN_ABLATE & 16) && it + 1 < ntile) stage((it & 1) ^ 1);
            if (!(VISREP_ATTN_ABLATE & 16)) __syncthreads();
        }
        return;
    }
    for (int it = 0; it < ntile; ++it) {
        const int cur = it & 1;
        if (!(VISREP_ATTN_ABLATE & 16) && it + 1 < ntile) stage(cur ^ 1);
        const char* sk = smem + ((VISREP_ATTN_ABLATE & 16) ? 0 : cur) * STAGE_B;
        const char* sv = sk + KV_B;

        // ---- S^T tiles: s[kt2] = K[kt2*32.., :] . Q^T
        f32x16 s[2];
        if (PS) { s[0] = minit; s[1] = minit; }
        else { s[0] = f32x16{}; s[1] = f32x16{}; }
#pragma unroll
        for (int kk = 0; kk < 4 * ND; ++kk) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2) {
                const bf16x8 kf = (VISREP_ATTN_ABLATE & 8) ? qf[(kk + kt2) % (4 * ND)]
                    : *reinterpret_cast<const bf16x8*>(sk + (kk >> 2) * TILE_B + kt2 * 4096 + rbase + (((2 * (kk & 3) + hi) ^ rsw) << 4));
                if (!(VISREP_ATTN_ABLATE & 4)) s[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[kt2], 0, 0, 0);
                else s[kt2][kk & 15] += (float)kf[0];
            }
        }
        // ---- mask (first / last tile of the image only), running max on the RAW scores; the softmax scale is folded
        //      into the exponent: p = exp2(s*sc - m*sc) is one FMA + one v_exp per element
        const int mt = m_begin + it * KT;
        if ((mt < tok0) || (mt + KT > tok1) || p.causal) {   // wave-uniform
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = mt + kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key < tok0 || key >= tok1 || (p.causal && key - tok0 > qloc)) s[kt2][r] = -INFINITY;
                }
        }
        float mloc = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]), mloc1 = fmaxf(fmaxf(s[1][0], s[1][1]), s[1][2]);   // two independent
#pragma unroll
        for (int r = 3; r < 15; r += 2) {                                                                         // v_max3_f32 chains
            mloc = fmaxf(fmaxf(mloc, s[0][r]), s[0][r + 1]);
            mloc1 = fmaxf(fmaxf(mloc1, s[1][r]), s[1][r + 1]);
        }
        mloc = fmaxf(fmaxf(mloc, s[0][15]), fmaxf(mloc1, s[1][15]));
        {   // the other half of this query's keys lives in lane ^ 32: one lane-swap VALU op (no LDS round trip)
            const unsigned u = __builtin_bit_cast(unsigned, mloc);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mloc = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
        }
        float psum = 0.f;
        uint32_t pb[2][8];
        if constexpr (PS) {
            // s = true exponent - m_run.  Move the reference on the first tile (to the row's true maximum, whatever its sign) and when a
            // row's maximum exceeds it by more than THR (p <= 2^THR otherwise: bf16 keeps its relative precision, sums are fp32)
            if (it == 0 || __any(mloc > (float)(VISREP_ATTN_V1_THR > 0 ? VISREP_ATTN_V1_THR : 0))) {     // wave-uniform
                const float delta = it == 0 ? mloc : fmaxf(mloc, 0.f);
#pragma unroll
                for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt2][r] -= delta;
                if (it > 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    l_run *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int dt = 0; dt < 2 * ND; ++dt) o[dt][r] *= alpha;
                }
                m_run += delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) minit[r] = -m_run;
            }
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = (VISREP_ATTN_ABLATE & 1) ? s[kt2][r] : __builtin_amdgcn_exp2f(s[kt2][r]);
                    const float p1 = (VISREP_ATTN_ABLATE & 1) ? s[kt2][r + 1] : __builtin_amdgcn_exp2f(s[kt2][r + 1]);
                    psum += p0 + p1;
                    pb[kt2][r >> 1] = pack_bf16(p0, p1);
                }
        } else {
#if VISREP_ATTN_V1_THR > 0                                  // deferred maximum: keep the old one while no row's maximum grew by more than THR (exp2 units):
        // p <= 2^THR then (bf16 keeps its relative precision, sums are fp32) and most tiles skip the rescale branch below; -2.4 % (round 3)
        const float m_new = __any((mloc - m_run) * p.sc > (float)VISREP_ATTN_V1_THR) ? fmaxf(m_run, mloc) : m_run;
#else
        const float m_new = fmaxf(m_run, mloc);             // finite: every image's first tile holds >= 1 valid key
#endif
        const float msc = m_new * p.sc;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float a0 = __builtin_fmaf(s[kt2][r], p.sc, -msc), a1 = __builtin_fmaf(s[kt2][r + 1], p.sc, -msc);
                const float p0 = (VISREP_ATTN_ABLATE & 1) ? a0 : __builtin_amdgcn_exp2f(a0);
                const float p1 = (VISREP_ATTN_ABLATE & 1) ? a1 : __builtin_amdgcn_exp2f(a1);
                psum += p0 + p1;
                pb[kt2][r >> 1] = pack_bf16(p0, p1);
            }
        if (__any(m_new > m_run)) {                          // wave-uniform: most tiles leave every row's max unchanged
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int dt = 0; dt < 2 * ND; ++dt) o[dt][r] *= alpha;
            m_run = m_new;
        }
        }
        l_run += psum;

        // ---- O^T += V^T P^T : chunk c = 16 keys; P operand = 4 packed words of s[c>>1], regs 8*(c&1) .. +8
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 pf;
            {
                u32x4 w = {pb[c >> 1][4 * (c & 1) + 0], pb[c >> 1][4 * (c & 1) + 1], pb[c >> 1][4 * (c & 1) + 2], pb[c >> 1][4 * (c & 1) + 3]};
                pf = *reinterpret_cast<bf16x8*>(&w);
            }
#pragma unroll
            for (int dt = 0; dt < 2 * ND; ++dt) {
                const bf16x8 vf = (VISREP_ATTN_ABLATE & 8) ? qf[(c + dt) % (4 * ND)]
                    : *reinterpret_cast<const bf16x8*>(sv + (dt >> 1) * TILE_B + (dt & 1) * 4096 + rbase + (((2 * c + hi) ^ rsw) << 4));
                if (!(VISREP_ATTN_ABLATE & 2)) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                else o[dt][c] += (float)vf[0] + (float)pf[1];
            }
        }
        if (!(VISREP_ATTN_ABLATE & 16)) __syncthreads();
    }

    // ---- normalise and store: lane holds O[q][d = dt*32 + 8*rg + 4*hi + (0..3)]
    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.f / l_run;
    // Two 8-column groups (rg, rg + 1) are regrouped with one v_permlane32_swap per word: lanes l and l + 32 hold the two halves of
    // every 8-column group, so after the swap the lower half-wave owns the 8 columns of group rg and the upper one those of rg + 1 ->
    // 16-byte stores, half as many (the store tail is issue-bound; all lanes take part in the swaps, only valid rows store).
    bf16_t* orow = p.out + ((size_t)b * p.Tq + (qloc < p.Tq ? qloc : p.Tq - 1)) * p.ldo + h * D;
    const bool wide = (p.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;          // uniform
#pragma unroll
    for (int dt = 0; dt < 2 * ND; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; rg += 2) {
            u32x2 a = {pack_bf16(o[dt][4 * rg + 0] * inv, o[dt][4 * rg + 1] * inv), pack_bf16(o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv)};
            u32x2 c = {pack_bf16(o[dt][4 * rg + 4] * inv, o[dt][4 * rg + 5] * inv), pack_bf16(o[dt][4 * rg + 6] * inv, o[dt][4 * rg + 7] * inv)};
            if (wide) {
                const auto w0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                const auto w1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                if (qloc < p.Tq) {
                    const u32x4 q4 = {(unsigned)w0[0], (unsigned)w1[0], (unsigned)w0[1], (unsigned)w1[1]};
                    *reinterpret_cast<u32x4*>(orow + dt * 32 + (rg + hi) * 8) = q4;
                }
            } else if (qloc < p.Tq) {
                *reinterpret_cast<u32x2*>(orow + dt * 32 + rg * 8 + hi * 4) = a;
                *reinterpret_cast<u32x2*>(orow + dt * 32 + (rg + 1) * 8 + hi * 4) = c;
            }
        }
}


// ---- image-aligned self-attention for CLS towers with (T - 1) % 64 == 0 (CLIP-L/14-336: 577 = 1 + 9 * 64; the 224-px towers: 257 = 1 + 4 * 64)
// and pre-scaled Q (round 4).  attn_fwd tiles the keys on GLOBAL 64-token blocks, so an image of 577 tokens always spans ten key tiles for
// 9.02 tiles of data, two of them masked.  Here the PATCH keys 1 .. T-1 of an image are exactly (T - 1) / 64 full tiles - K rows read from
// row b T + 1 on, V^T columns image-aligned (the V projection writes them that way: a row-mapped GEMM over the patch tokens, column
// b (T - 1) + t - 1) - and the CLS key is a rank-1 side term handled once per wave before the tile loop:
//     s_cls[q] = q . k_cls (32 FMAs per lane + one lane swap), reference m = s_cls, p_cls = 1, l = 1, O = v_cls
// (v_cls: the V projection of the CLS rows, [B, H * 64] row-major).  No masks, no first / last tile special cases, one tile less per image.
struct AttnClsArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* vt; const bf16_t* vcls; bf16_t* out;
    int B, T, H, ldq, ldk, ldvt, ldvc, ldo;
};

__global__ __launch_bounds__(256, 4) void attn_fwd_cls(const AttnClsArgs p) {
    constexpr int STAGE_B = 2 * TILE_B;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const int nqt = (p.T + 127) >> 7, P = p.T - 1, ntile = P >> 6;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int h = id % p.H;
    const int b = id / p.H;

    const int qloc = qt * 128 + wave * 32 + lq;
    const size_t qrow = (size_t)b * p.T + (qloc < p.T ? qloc : p.T - 1);
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(p.q + qrow * p.ldq + h * 64 + kk * 16 + hi * 8);

    // ---- staging: every tile is full and in range, the cursors only advance
    const int srow = tid >> 3;
    const int lslot = (tid & 7) ^ ((srow >> 1) & 7);
    const bf16_t* kcur = p.k + h * 64 + lslot * 8 + ((size_t)b * p.T + 1 + srow) * p.ldk;
    const bf16_t* vcur = p.vt + (size_t)(h * 64 + srow) * p.ldvt + lslot * 8 + (size_t)b * P;
    const size_t kstep = (size_t)KT * p.ldk, khalf = (size_t)32 * p.ldk, vhalf = (size_t)32 * p.ldvt;
    auto stage = [&](int buf) {
        char* sk = smem + buf * STAGE_B + wave * 1024;
        char* sv = sk + TILE_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(kcur + j * khalf, sk + j * 4096);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(vcur + j * vhalf, sv + j * 4096);
        kcur += kstep; vcur += KT;
    };
    const int rsw = (lq >> 1) & 7;
    const int rbase = lq * 128;

    stage(0);
    __syncthreads();
    if (qt * 128 + wave * 32 >= p.T) {                         // an empty wave only stages its share and meets the barriers
        for (int it = 0; it < ntile; ++it) {
            if (it + 1 < ntile) stage((it & 1) ^ 1);
            __syncthreads();
        }
        return;
    }

    // ---- the CLS key: s_cls = q . k_cls over this lane's half of the head (d = 16 kk + 8 hi .. + 8), the other half lives in lane ^ 32
    float m_run, l_run;
    f32x16 o[2], minit;
    {
        const bf16_t* kc = p.k + (size_t)b * p.T * p.ldk + h * 64 + hi * 8;
        float dot = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const u32x4 kw = *reinterpret_cast<const u32x4*>(kc + kk * 16);
            const u32x4 qw = __builtin_bit_cast(u32x4, qf[kk]);
#pragma unroll
            for (int e = 0; e < 4; ++e) dot = __builtin_fmaf(bf_lo(qw[e]), bf_lo(kw[e]), __builtin_fmaf(bf_hi(qw[e]), bf_hi(kw[e]), dot));
        }
        dot += __shfl_xor(dot, 32);
        m_run = dot;                                          // the reference: p_cls = 2^0
        l_run = hi == 0 ? 1.f : 0.f;                          // the two halves of a row are added at the end: count the CLS key once
        // O = p_cls * v_cls: lane holds O[q][d = 32 dt + 8 rg + 4 hi + e] in o[dt][4 rg + e]
        const bf16_t* vc = p.vcls + (size_t)b * p.ldvc + h * 64 + hi * 4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const u32x2 vw = *reinterpret_cast<const u32x2*>(vc + dt * 32 + rg * 8);
                o[dt][4 * rg + 0] = bf_lo(vw[0]); o[dt][4 * rg + 1] = bf_hi(vw[0]);
                o[dt][4 * rg + 2] = bf_lo(vw[1]); o[dt][4 * rg + 3] = bf_hi(vw[1]);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) minit[r] = -m_run;
    }

    for (int it = 0; it < ntile; ++it) {
        const int cur = it & 1;
        if (it + 1 < ntile) stage(cur ^ 1);
        const char* sk = smem + cur * STAGE_B;
        const char* sv = sk + TILE_B;
        f32x16 s[2];
        s[0] = minit; s[1] = minit;                           // the reference is subtracted inside the matrix pipe
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + kt2 * 4096 + rbase + (((2 * kk + hi) ^ rsw) << 4));
                s[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[kt2], 0, 0, 0);
            }
        float mloc = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]), mloc1 = fmaxf(fmaxf(s[1][0], s[1][1]), s[1][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) {
            mloc = fmaxf(fmaxf(mloc, s[0][r]), s[0][r + 1]);
            mloc1 = fmaxf(fmaxf(mloc1, s[1][r]), s[1][r + 1]);
        }
        mloc = fmaxf(fmaxf(mloc, s[0][15]), fmaxf(mloc1, s[1][15]));
        {
            const unsigned u = __builtin_bit_cast(unsigned, mloc);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mloc = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
        }
        if (__any(mloc > (float)(VISREP_ATTN_V1_THR > 0 ? VISREP_ATTN_V1_THR : 0))) {     // wave-uniform: some row outgrew the reference by > 2^THR
            const float delta = fmaxf(mloc, 0.f);
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt2][r] -= delta;
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            m_run += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) minit[r] = -m_run;
        }
        float psum = 0.f;
        uint32_t pb[2][8];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(s[kt2][r]), p1 = __builtin_amdgcn_exp2f(s[kt2][r + 1]);
                psum += p0 + p1;
                pb[kt2][r >> 1] = pack_bf16(p0, p1);
            }
        l_run += psum;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4 w = {pb[c >> 1][4 * (c & 1) + 0], pb[c >> 1][4 * (c & 1) + 1], pb[c >> 1][4 * (c & 1) + 2], pb[c >> 1][4 * (c & 1) + 3]};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, w);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sv + dt * 4096 + rbase + (((2 * c + hi) ^ rsw) << 4));
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.f / l_run;
    bf16_t* orow = p.out + ((size_t)b * p.T + (qloc < p.T ? qloc : p.T - 1)) * p.ldo + h * 64;
    const bool wide = (p.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;          // uniform
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; rg += 2) {
            u32x2 a = {pack_bf16(o[dt][4 * rg + 0] * inv, o[dt][4 * rg + 1] * inv), pack_bf16(o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv)};
            u32x2 c = {pack_bf16(o[dt][4 * rg + 4] * inv, o[dt][4 * rg + 5] * inv), pack_bf16(o[dt][4 * rg + 6] * inv, o[dt][4 * rg + 7] * inv)};
            if (wide) {
                const auto w0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                const auto w1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                if (qloc < p.T) {
                    const u32x4 q4 = {(unsigned)w0[0], (unsigned)w1[0], (unsigned)w0[1], (unsigned)w1[1]};
                    *reinterpret_cast<u32x4*>(orow + dt * 32 + (rg + hi) * 8) = q4;
                }
            } else if (qloc < p.T) {
                *reinterpret_cast<u32x2*>(orow + dt * 32 + rg * 8 + hi * 4) = a;
                *reinterpret_cast<u32x2*>(orow + dt * 32 + (rg + 1) * 8 + hi * 4) = c;
            }
        }
}


// ---- wide-head attention (head width D = 64 ND up to 512: the single 512-wide head of the diffusion VAE's mid-block attention over all 9216
// latent pixels, AutoencoderKL encoder).  The flash loop of attn_fwd with the register file of ONE wave per SIMD: 32 Q fragments (128 VGPRs) and
// the 32 x 512 output accumulators (256 registers) stay resident; a K tile and a V^T tile (64 keys x 512 = 64 KB each) are single-buffered
// and refilled alternately - K(t+1) streams in under tile t's softmax and P.V, V^T(t+1) under tile t+1's Q.K^T - with counted vmcnt waits
// (the LDS-DMA queue is in order: K(t+1) was issued before V^T(t+1), V^T(t) before K(t+1)).  Fragment reads are hand-written ds_read_b128
// (hipcc would otherwise put an s_waitcnt vmcnt(0) in front of every LDS read that follows an LDS-DMA issue and serialise the refills).
// Key tiles must be whole (Tk % 64 == 0): no masks.  Replaces the materialised-score route (fp32 [T, T] scores per image through the GEMM
// kernel, softmax_rows, P.V GEMM: ~1 GB of HBM traffic per image at T = 9216).
VR_DEV unsigned attn_lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
VR_DEV void lds_read4(bf16x8 (&f)[4], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}
VR_DEV void lds_wait4(bf16x8 (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); }

template <int ND, int NDO>                  // NDO = 64-wide blocks of the head a workgroup accumulates O for (the register budget: 32 x 64 NDO fp32 per wave)
__global__ __launch_bounds__(256, 1) void attn_fwd_wide(const AttnArgs p) {
    constexpr int D = 64 * ND, KV_B = ND * TILE_B, NPK = 2 * ND, NPV = 2 * NDO;      // LDS-DMA pieces per wave: K tile / V^T tile
    static_assert(NPK <= 31 && ND % NDO == 0, "the counted waits below encode the piece counts in vmcnt");
    extern __shared__ __attribute__((aligned(16))) char smem[];           // K tile | V^T tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const int nqt = (p.Tq + 127) >> 7;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int dh = id % (ND / NDO); id /= (ND / NDO);          // which slice of the head's width this workgroup produces (Q.K^T is over all of it)
    const int qt = id % nqt; id /= nqt;
    const int h = id % p.H;
    const int b = id / p.H;
    const int tok0 = p.kv_shared ? 0 : b * p.Tk;
    const int ntile = p.Tk >> 6;

    const int qloc = qt * 128 + wave * 32 + lq;
    const size_t qrow = (size_t)b * p.Tq + (qloc < p.Tq ? qloc : p.Tq - 1);
    bf16x8 qf[4 * ND];
#pragma unroll
    for (int kk = 0; kk < 4 * ND; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(p.q + qrow * p.ldq + h * D + kk * 16 + hi * 8);

    // Staging addresses = a wave-uniform tile / piece base (scalar registers, scalar arithmetic) + ONE 32-bit per-lane byte offset per operand:
    // with a 64-bit per-lane pointer per piece the 24 pieces of a tile pair cost 48 registers of loop-invariant addresses (spilled).
    const int srow = tid >> 3;
    const int lslot = (tid & 7) ^ ((srow >> 1) & 7);
    const unsigned koff = (unsigned)(srow * p.ldk + lslot * 8) * 2u, voff = (unsigned)(srow * p.ldvt + lslot * 8) * 2u;
    const char* const kbase_g = reinterpret_cast<const char*>(p.k + h * D + (size_t)tok0 * p.ldk);
    const char* const vbase_g = reinterpret_cast<const char*>(p.vt + (size_t)(h * D + dh * NDO * 64) * p.ldvt + tok0);
    const size_t kstep = (size_t)KT * p.ldk * 2, khalf = (size_t)32 * p.ldk * 2, vhalf = (size_t)32 * p.ldvt * 2;
    char* const sk_w = smem + wave * 1024;
    char* const sv_w = smem + KV_B + wave * 1024;
    int kt_next = 0, vt_next = 0;                              // next tile to stage (uniform)
    auto stage_k = [&]() {
        const char* const t = kbase_g + (size_t)kt_next * kstep;
        unsigned ko = koff;
        asm volatile("" : "+v"(ko));                            // opaque per call: keeps "piece base + lane offset" a two-instruction add per piece
#pragma unroll                                                  // instead of 16 hoisted (and spilled) 64-bit per-lane addresses
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(t + j * khalf + nd * 128 + (size_t)ko, sk_w + nd * TILE_B + j * 4096);
        ++kt_next;
    };
    auto stage_v = [&]() {
        const char* const t = vbase_g + (size_t)vt_next * (KT * 2);
        unsigned vo = voff;
        asm volatile("" : "+v"(vo));
#pragma unroll
        for (int nd = 0; nd < NDO; ++nd)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(t + (size_t)(2 * nd + j) * vhalf + (size_t)vo, sv_w + nd * TILE_B + j * 4096);
        ++vt_next;
    };
    auto wait_k = [&]() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPV) : "memory"); };     // K(t) landed: only the V^T pieces issued after it may be pending
    auto wait_v = [&]() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPK) : "memory"); };     // V^T(t) landed: only K(t + 1) may be pending
    auto wait_0 = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    const int rsw = (lq >> 1) & 7;
    const unsigned kbase = attn_lds_addr(smem) + lq * 128, vbase = kbase + KV_B;
    unsigned foff[8];                                          // byte offset of logical 16-byte slot 2 c + hi (c = 0..3), + 4096 for the second 32 rows
#pragma unroll
    for (int c = 0; c < 4; ++c) { foff[c] = ((2 * c + hi) ^ rsw) << 4; foff[4 + c] = foff[c] + 4096; }

    // Softmax reference.  O is never rescaled inside the tile loop: a VALU multiply of the accumulators in the loop body makes the register
    // allocator move all of O (128 accumulation registers) into the architectural file at every loop entry, which evicts the Q fragments to
    // scratch.  Instead every row's exponentials are taken against a FIXED reference m_ref - the row's maximum over the first key tile, a lower
    // bound of the row's true maximum - so p = 2^((s - m_ref) sc) >= 1 for the largest key (no underflow of the sums) and fp32 / bf16 hold it up
    // to 2^127: only a row whose later scores exceed its first tile's by more than 2^100 could overflow.  The true running maximum is tracked on
    // the side; if any row of the workgroup crossed that bound the whole pass is repeated once with m_ref = the true maximum (then p <= 1).
    f32x16 o[2 * NDO];
    float m_ref = 0.f, m_true = 0.f, l_run = 0.f;
    for (int attempt = 0; attempt < 2; ++attempt) {
#pragma unroll
    for (int dt = 0; dt < 2 * NDO; ++dt) o[dt] = f32x16{};
    l_run = 0.f;
    kt_next = 0; vt_next = 0;
    stage_k();
    stage_v();
    for (int it = 0; it < ntile; ++it) {
        const bool more = it + 1 < ntile;                      // uniform
        // ---- K(it) has landed everywhere (V^T(it), issued after it, may still be in flight)
        wait_k();
        bar();
        f32x16 s[2];
        s[0] = f32x16{}; s[1] = f32x16{};
        {
            bf16x8 fa[4], fb[4];
            // S^T += K[kt2 * 32 + lq][16 kk + 8 hi ..] . Q: groups of four fragment reads = (kk, kk + 1) x (kt2 = 0, 1), one group ahead
            lds_read4(fa, kbase + foff[0], kbase + foff[4], kbase + foff[1], kbase + foff[5]);
#pragma unroll
            for (int g = 0; g < 2 * ND; ++g) {                 // group g: kk = 2 g, 2 g + 1
                bf16x8 (&cur)[4] = (g & 1) ? fb : fa;
                bf16x8 (&nxt)[4] = (g & 1) ? fa : fb;
                lds_wait4(cur);
                if (g + 1 < 2 * ND) {
                    const int k2 = 2 * (g + 1);
                    const unsigned t = kbase + (k2 >> 2) * TILE_B;
                    lds_read4(nxt, t + foff[k2 & 3], t + foff[4 + (k2 & 3)], t + foff[(k2 + 1) & 3], t + foff[4 + ((k2 + 1) & 3)]);
                }
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[0], qf[2 * g], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[1], qf[2 * g], s[1], 0, 0, 0);
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[2], qf[2 * g + 1], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[3], qf[2 * g + 1], s[1], 0, 0, 0);
            }
        }
        // ---- every wave is done with the K tile: refill it (under the softmax and P.V of this tile)
        bar();
        if (more) stage_k();
        uint32_t pb[2][8];
        {
            float mloc = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]), mloc1 = fmaxf(fmaxf(s[1][0], s[1][1]), s[1][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) {
                mloc = fmaxf(fmaxf(mloc, s[0][r]), s[0][r + 1]);
                mloc1 = fmaxf(fmaxf(mloc1, s[1][r]), s[1][r + 1]);
            }
            mloc = fmaxf(fmaxf(mloc, s[0][15]), fmaxf(mloc1, s[1][15]));
            {
                const unsigned u = __builtin_bit_cast(unsigned, mloc);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mloc = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
            if (it == 0) {                                      // uniform
                if (attempt == 0) m_ref = mloc;
                m_true = mloc;
            } else {
                m_true = fmaxf(m_true, mloc);
            }
            const float msc = m_ref * p.sc;
            float psum = 0.f;
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt2][r], p.sc, -msc));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt2][r + 1], p.sc, -msc));
                    psum += p0 + p1;
                    pb[kt2][r >> 1] = pack_bf16(p0, p1);
                }
            l_run += psum;
        }
        // ---- V^T(it) has landed everywhere (K(it + 1) may be in flight)
        if (more) wait_v(); else wait_0();
        bar();
        {
            bf16x8 pf[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 w = {pb[c >> 1][4 * (c & 1) + 0], pb[c >> 1][4 * (c & 1) + 1], pb[c >> 1][4 * (c & 1) + 2], pb[c >> 1][4 * (c & 1) + 3]};
                pf[c] = __builtin_bit_cast(bf16x8, w);
            }
            // O^T[dt] += V^T[dt * 32 + lq][16 c + ..] . P: one group = the four key chunks c of one 32-row block dt
            bf16x8 fa[4], fb[4];
            lds_read4(fa, vbase + foff[0], vbase + foff[1], vbase + foff[2], vbase + foff[3]);
#pragma unroll
            for (int dt = 0; dt < 2 * NDO; ++dt) {
                bf16x8 (&cur)[4] = (dt & 1) ? fb : fa;
                bf16x8 (&nxt)[4] = (dt & 1) ? fa : fb;
                lds_wait4(cur);
                if (dt + 1 < 2 * NDO) {
                    const unsigned t = vbase + ((dt + 1) >> 1) * TILE_B + ((dt + 1) & 1) * 4096;
                    lds_read4(nxt, t + foff[0], t + foff[1], t + foff[2], t + foff[3]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[c], pf[c], o[dt], 0, 0, 0);
            }
        }
        // ---- every wave is done with the V^T tile: refill it (under the next tile's Q.K^T)
        bar();
        if (more) stage_v();
    }
    // ---- did any row of the workgroup outgrow its reference by more than 2^100 ?  (block-uniform: the waves share the tiles and the barriers)
    const bool over = __any((m_true - m_ref) * p.sc > 100.f) != 0;
    int* const flags = reinterpret_cast<int*>(smem);            // the K tile is idle: every wave passed the loop's last barrier, nothing is in flight
    if (lane == 0) flags[wave] = over ? 1 : 0;
    __syncthreads();
    const int any_over = flags[0] | flags[1] | flags[2] | flags[3];
    __syncthreads();                                            // before anybody's LDS-DMA overwrites the flags
    if (!any_over) break;
    m_ref = m_true;
    }

    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.f / l_run;
    bf16_t* orow = p.out + ((size_t)b * p.Tq + (qloc < p.Tq ? qloc : p.Tq - 1)) * p.ldo + h * D + dh * NDO * 64;
    const bool wide = (p.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;          // uniform
#pragma unroll
    for (int dt = 0; dt < 2 * NDO; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; rg += 2) {
            u32x2 a = {pack_bf16(o[dt][4 * rg + 0] * inv, o[dt][4 * rg + 1] * inv), pack_bf16(o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv)};
            u32x2 c = {pack_bf16(o[dt][4 * rg + 4] * inv, o[dt][4 * rg + 5] * inv), pack_bf16(o[dt][4 * rg + 6] * inv, o[dt][4 * rg + 7] * inv)};
            if (wide) {
                const auto w0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                const auto w1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                if (qloc < p.Tq) {
                    const u32x4 q4 = {(unsigned)w0[0], (unsigned)w1[0], (unsigned)w0[1], (unsigned)w1[1]};
                    *reinterpret_cast<u32x4*>(orow + dt * 32 + (rg + hi) * 8) = q4;
                }
            } else if (qloc < p.Tq) {
                *reinterpret_cast<u32x2*>(orow + dt * 32 + rg * 8 + hi * 4) = a;
                *reinterpret_cast<u32x2*>(orow + dt * 32 + (rg + 1) * 8 + hi * 4) = c;
            }
        }
}

}  // namespace

thread_local int t_visrep_attn_variant = 1;   // 1 = attn_fwd<ND> for every head width (default); 2 = attn_fwd_ab (attention_ab.hip, VISREP_EXPERIMENTS builds) for head width 64

extern "C" int visrep_set_attn_variant(int variant) {          // per-thread; returns the previous value
#ifdef VISREP_EXPERIMENTS
    const bool ok = variant == 1 || variant == 2;
#else
    const bool ok = variant == 1;
#endif
    if (!ok) return visrep_set_error(VISREP_ERR_SHAPE, "attention variant must be 1 (2 = attn_fwd_ab: VISREP_EXPERIMENTS builds only)");
    const int old = t_visrep_attn_variant;
    t_visrep_attn_variant = variant;
    return old;
}

extern "C" int visrep_attention_fwd(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* out, int ldo,
                                    int B, int Tq, int Tk, int H, int head_dim, int kv_shared, int causal, float scale, void* stream) {
    if (head_dim != 64 && head_dim != 128 && head_dim != 192 && head_dim != 512)
        return visrep_set_error(VISREP_ERR_SHAPE, "attention: head_dim must be 64, 128, 192 or 512 (pad narrower heads with zero weights)");
    if (B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "attention: empty problem");
    if ((ldq % 8) || (ldk % 8) || (ldvt % 64) || (ldo % 4)) return visrep_set_error(VISREP_ERR_SHAPE, "attention: bad leading dimension");
    const long Mk = kv_shared ? Tk : (long)B * Tk;
    if (ldvt < ((Mk + 63) / 64) * 64) return visrep_set_error(VISREP_ERR_SHAPE, "attention: ldvt must cover round_up(key rows, 64)");
    if (head_dim == 512) {                                      // attn_fwd_wide<8, 4>: whole key tiles, no masks; two workgroups per query tile (256 output columns each)
        if (causal || Tk % 64 || scale <= 0.f) return visrep_set_error(VISREP_ERR_SHAPE, "attention: head_dim 512 needs Tk % 64 == 0, no causal mask, scale > 0");
        AttnArgs w;
        w.q = (const bf16_t*)q; w.k = (const bf16_t*)k; w.vt = (const bf16_t*)vt; w.out = (bf16_t*)out;
        w.B = B; w.Tq = Tq; w.Tk = Tk; w.H = H; w.Mk = (int)Mk; w.ldq = ldq; w.ldk = ldk; w.ldvt = ldvt; w.ldo = ldo; w.kv_shared = kv_shared; w.causal = 0;
        w.sc = scale * 1.4426950408889634f;
        static VisrepLdsOptIn optw;
        visrep_lds_opt_in(optw, (const void*)attn_fwd_wide<8, 4>, 16 * TILE_B);
        visrep_count_route(VISREP_ROUTE_ATTN_WIDE);
        hipLaunchKernelGGL((attn_fwd_wide<8, 4>), dim3(2 * ((Tq + 127) / 128) * H * B), dim3(256), (size_t)16 * TILE_B, (hipStream_t)stream, w);
        return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "attention: launch failed");
    }
#ifdef VISREP_EXPERIMENTS
    if (head_dim == 64 && t_visrep_attn_variant == 2)
        return visrep_attention_ab_launch(q, ldq, k, ldk, vt, ldvt, out, ldo, B, Tq, Tk, H, kv_shared, causal, scale, (hipStream_t)stream);
#endif
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out;
    a.B = B; a.Tq = Tq; a.Tk = Tk; a.H = H; a.Mk = (int)Mk; a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo; a.kv_shared = kv_shared; a.causal = causal;
    a.sc = scale * 1.4426950408889634f;
    const bool ps = scale <= 0.f;                           // the caller folded scale * log2(e) into Q (see attn_fwd PS)
    if (ps && head_dim != 64) return visrep_set_error(VISREP_ERR_ARG, "attention: pre-scaled Q (scale <= 0) is built for head_dim 64");
    const int nqt = (Tq + 127) / 128, nd = head_dim / 64;
    const dim3 grid(nqt * H * B), block(256);
    size_t lds = (size_t)nd * 4 * TILE_B;                   // double-buffered K + V^T tiles
#ifdef VISREP_EXPERIMENTS
    {   // diagnostic (tools-only build): unused LDS to lower the number of resident workgroups per CU; the knob is read once
        static const int extra = getenv("VISREP_ATTN_EXTRA_LDS") ? atoi(getenv("VISREP_ATTN_EXTRA_LDS")) : 0;
        if (extra > 0 && nd == 1) {
            lds += (size_t)extra;
            static VisrepLdsOptIn opt1;
            visrep_lds_opt_in(opt1, (const void*)attn_fwd<1>, 160 * 1024);
        }
    }
#endif
    hipStream_t st = (hipStream_t)stream;
    visrep_count_route(VISREP_ROUTE_ATTN);
    if (nd == 1 && ps) hipLaunchKernelGGL((attn_fwd<1, true>), grid, block, lds, st, a);
    else if (nd == 1) hipLaunchKernelGGL(attn_fwd<1>, grid, block, lds, st, a);
    else if (nd == 2) hipLaunchKernelGGL(attn_fwd<2>, grid, block, lds, st, a);
    else {
        static VisrepLdsOptIn opt3;                         // 96 KB of dynamic LDS needs the opt-in once per device
        visrep_lds_opt_in(opt3, (const void*)attn_fwd<3>, (int)lds);
        hipLaunchKernelGGL(attn_fwd<3>, grid, block, lds, st, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "attention: launch failed");
}

extern "C" int visrep_mhsa_cls_supported(int T) { return T > 64 && (T - 1) % 64 == 0 ? 1 : 0; }

// Image-aligned self-attention of a CLS tower with pre-scaled Q (attn_fwd_cls): qk as for visrep_mhsa_fwd; vt = V^T of the PATCH tokens,
// column b (T - 1) + perm16(t - 1) (a row-mapped V projection writes it); vcls = the V projection of the CLS rows [B, ldvc].
extern "C" int visrep_mhsa_cls_fwd(const void* qk, int ldqk, const void* vt, int ldvt, const void* vcls, int ldvc, void* out, int ldo, int B, int T, int H,
                                   int head_dim, void* stream) {
    if (head_dim != 64) return visrep_set_error(VISREP_ERR_SHAPE, "mhsa_cls: only head_dim 64 is implemented");
    if (!qk || !vt || !vcls || !out) return visrep_set_error(VISREP_ERR_ARG, "mhsa_cls: null pointer");
    if (B <= 0 || H <= 0 || !visrep_mhsa_cls_supported(T)) return visrep_set_error(VISREP_ERR_SHAPE, "mhsa_cls: needs T = 1 + a multiple of 64 (CLS + full key tiles)");
    if ((ldqk % 8) || (ldvt % 64) || (ldo % 4) || (ldvc % 4)) return visrep_set_error(VISREP_ERR_SHAPE, "attention: bad leading dimension");
    if (ldvt < B * (T - 1)) return visrep_set_error(VISREP_ERR_SHAPE, "mhsa_cls: ldvt must cover B (T - 1) columns");
    AttnClsArgs a;
    a.q = (const bf16_t*)qk; a.k = (const bf16_t*)qk + (size_t)H * 64; a.vt = (const bf16_t*)vt; a.vcls = (const bf16_t*)vcls; a.out = (bf16_t*)out;
    a.B = B; a.T = T; a.H = H; a.ldq = ldqk; a.ldk = ldqk; a.ldvt = ldvt; a.ldvc = ldvc; a.ldo = ldo;
    visrep_count_route(VISREP_ROUTE_ATTN_CLS);
    hipLaunchKernelGGL(attn_fwd_cls, dim3(((T + 127) / 128) * H * B), dim3(256), (size_t)4 * TILE_B, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "mhsa_cls: launch failed");
}

extern "C" int visrep_mhsa_fwd(const void* qk, int ldqk, const void* vt, int ldvt, void* out, int ldo,
                               int B, int T, int H, int head_dim, float scale, void* stream) {
    if (head_dim != 64) return visrep_set_error(VISREP_ERR_SHAPE, "mhsa: only head_dim 64 is implemented");
    return visrep_attention_fwd(qk, ldqk, (const bf16_t*)qk + (size_t)H * 64, ldqk, vt, ldvt, out, ldo, B, T, T, H, 64, 0, 0, scale, stream);
}
