// A score on gfx950:  score[img] = mean_t max_s cos(other[img][t], ref[img][s])
// (reference: A_score/compute.py:12-15 normalize_feat, :54-72 cosine_similarity -> max(dim=1) -> mean).
//
// The reference materialises an [Nt, Nr, D] broadcast product (5.4 GB at 576x576x4096); here the cross-Gram is an
// MFMA contraction whose [Nt, Nr] result never leaves registers: per-row scale factors (the two normalisations of the
// reference folded into one fp32 factor per row) are applied in the epilogue, the row max over s is taken in the
// accumulator layout and only one partial sum per 64-row tile is written.
//
//   pass 1  ascore_row_scale : c(x) = 1/(|x|+1e-10) / max(|x|/(|x|+1e-10), 1e-8)   one wave per row, 16-B loads
//   pass 2  ascore_maxcos    : "swapped" Gram  G^T = R O^T  (A operand = ref rows s, B operand = other rows t), so a
//           lane owns ONE target row t = lane&31 and sees 16 s per MFMA tile -> max over s is in-lane + one lane^32
//           exchange.  Operands are loaded straight from L2/HBM, 16 B per lane per row (K index = summation index:
//           both operands use the same lane->k map, so any k permutation is legal):
//             bf16 inputs: v_mfma_f32_32x32x16_bf16 (products exact in fp32, fp32 accumulate)
//             fp32 inputs: 4 x v_mfma_f32_32x32x2_f32 per 16-B load (exact fp32 FMA chain)
//   pass 3  ascore_finalize  : score[img] = sum(partials) / Nt   (fixed order -> deterministic)
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "visrep_internal.h"

constexpr int ASCORE_DEEP = 1;     // six-slot ring for 192 x 192 tiles (576 x 576 x 4096: 0.720 -> 0.651 ms with it, profiles/round3_scores_kernel_stats.md)
thread_local int t_visrep_ascore_variant = 0;   // 0 = pick by launched tile area, 1 = 128 x 128 tiles, 2 = persistent ping-pong tiles (visrep_set_ascore_variant)

namespace {

template <typename T>
__global__ __launch_bounds__(256) void ascore_row_scale(const T* __restrict__ x, long rows, int D, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * D;
    float sq = 0.f;
    if (sizeof(T) == 2) {
        for (int c = lane * 8; c < D; c += 64 * 8) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float a = bf_lo(v[k]), b = bf_hi(v[k]); sq += a * a + b * b; }
        }
    } else {
        for (int c = lane * 4; c < D; c += 64 * 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    }
    sq = wave_sum(sq);
    if (lane == 0) {
        const float n = sqrtf(sq);
        const float a = n + 1e-10f;                    // normalize_feat epsilon (compute.py:12-15)
        scale[row] = (1.0f / a) / fmaxf(n / a, 1e-8f);   // F.cosine_similarity eps on the already-normalised row
    }
}

struct AScoreArgs {
    const void* other; const void* ref;
    const float* c_other; const float* c_ref;
    float* partial;
    int n_img, Nt, Nr, D;
};

template <typename T>
__global__ __launch_bounds__(256) void ascore_maxcos(const AScoreArgs p) {
    __shared__ float red[4][32];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const int ntt = (p.Nt + 63) >> 6;
    const int img = blockIdx.x / ntt, tt = blockIdx.x - img * ntt;
    const int wt = wave & 1, ws = wave >> 1;
    const T* other = reinterpret_cast<const T*>(p.other) + (size_t)img * p.Nt * p.D;
    const T* ref = reinterpret_cast<const T*>(p.ref) + (size_t)img * p.Nr * p.D;
    const float* cr = p.c_ref + (size_t)img * p.Nr;
    const int t = tt * 64 + wt * 32 + lq;
    const int tc = t < p.Nt ? t : p.Nt - 1;
    constexpr int EPL = 16 / sizeof(T);                     // elements per 16-B load: 8 bf16 / 4 fp32
    const T* trow = other + (size_t)tc * p.D + hi * EPL;
    const int nst = (p.Nr + 63) >> 6;
    float best = -INFINITY;
    for (int st = ws; st < nst; st += 2) {
        const int s_a = st * 64 + lq, s_b = s_a + 32;
        const T* r0 = ref + (size_t)(s_a < p.Nr ? s_a : p.Nr - 1) * p.D + hi * EPL;
        const T* r1 = ref + (size_t)(s_b < p.Nr ? s_b : p.Nr - 1) * p.D + hi * EPL;
        f32x16 acc0 = f32x16{}, acc1 = f32x16{};
        if (sizeof(T) == 2) {
            int k = 0;
            for (; k + 64 <= p.D; k += 64) {             // 12 independent 16-B loads in flight per lane
                bf16x8 tb[4], a0[4], a1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    tb[u] = *reinterpret_cast<const bf16x8*>(trow + k + 16 * u);
                    a0[u] = *reinterpret_cast<const bf16x8*>(r0 + k + 16 * u);
                    a1[u] = *reinterpret_cast<const bf16x8*>(r1 + k + 16 * u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[u], tb[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[u], tb[u], acc1, 0, 0, 0);
                }
            }
            for (; k < p.D; k += 16) {
                const bf16x8 tb = *reinterpret_cast<const bf16x8*>(trow + k);
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(r0 + k);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(r1 + k);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, tb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, tb, acc1, 0, 0, 0);
            }
        } else {
            auto step = [&](const float4& tb, const float4& a0, const float4& a1) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, tb.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, tb.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, tb.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, tb.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, tb.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, tb.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, tb.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, tb.w, acc1, 0, 0, 0);
            };
            int k = 0;
            for (; k + 16 <= p.D; k += 16) {
                const float4 tb0 = *reinterpret_cast<const float4*>(trow + k), tb1 = *reinterpret_cast<const float4*>(trow + k + 8);
                const float4 a00 = *reinterpret_cast<const float4*>(r0 + k), a01 = *reinterpret_cast<const float4*>(r0 + k + 8);
                const float4 a10 = *reinterpret_cast<const float4*>(r1 + k), a11 = *reinterpret_cast<const float4*>(r1 + k + 8);
                step(tb0, a00, a10);
                step(tb1, a01, a11);
            }
            for (; k < p.D; k += 8)
                step(*reinterpret_cast<const float4*>(trow + k), *reinterpret_cast<const float4*>(r0 + k),
                     *reinterpret_cast<const float4*>(r1 + k));
        }
        // lane holds G[s = st*64 + blk*32 + row(r,hi)][t]; scale by c_ref[s], mask s >= Nr, running max
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s0 = st * 64 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (s0 < p.Nr) best = fmaxf(best, acc0[r] * cr[s0]);
            if (s0 + 32 < p.Nr) best = fmaxf(best, acc1[r] * cr[s0 + 32]);
        }
    }
    best = fmaxf(best, __shfl_xor(best, 32));
    if (hi == 0) red[wave][lq] = best;
    __syncthreads();
    if (wave < 2) {                                         // wave == wt here; combine the two s-halves, scale by c_other[t]
        float v = 0.f;
        if (hi == 0 && t < p.Nt) v = fmaxf(red[wave][lq], red[wave + 2][lq]) * p.c_other[(size_t)img * p.Nt + t];
        v = wave_sum(v);
        if (lane == 0) red[wave][0] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) p.partial[blockIdx.x] = red[0][0] + red[1][0];
}

__global__ void ascore_finalize(const float* __restrict__ partial, float* __restrict__ score, int n_img, int ntt, int Nt) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= n_img) return;
    float s = 0.f;
    for (int i = 0; i < ntt; ++i) s += partial[img * ntt + i];
    score[img] = s / (float)Nt;
}

// ---- bf16 production path: LDS-staged MFMA Gram (same structure and LDS tile format as gemm_bf16.hip v1) --------------
// One workgroup = ONE 128 x 128 tile of one image's [Nt, Nr] similarity: it sweeps all of D with double-buffered global_load_lds
// staging, scales the columns by c_ref, takes the row max over its 128 reference rows and writes it to rowmax[img][t][ref tile];
// ascore_finalize_tiles then takes the max over the reference tiles, applies c_other and averages.  The [Nt, Nr] similarity
// never exists in memory.
// Work mapping (round 2, from the PMC pass in profiles/round2_pmc_midround.md): round 1 gave a workgroup one 128-row target tile
// and looped over the reference tiles, re-streaming its target tile per reference tile and the whole reference per workgroup;
// one image's operands (9.4 MB at 576 x 576 x 4096) do not fit an XCD's 4 MB L2, the L2 hit rate was 14-29 % and the kernel ran at
// 7.1 TB/s of FABRIC traffic - 2.5x its algorithmic bytes: HBM-bound on re-fetches.  Now all ntt x nnt tile pairs of an image are
// consecutive logical workgroups (xcd_remap puts consecutive ids on one XCD): they start together, walk D in step, and at any
// moment the XCD's L2 only has to hold the current D-window of a few images, so every operand slab is fetched from HBM once and
// hit by the other 4-5 workgroups that need it.  (Also tried: taking the row norms from the MFMA fragments inside this kernel instead
// of the separate ascore_row_scale pass - the unpack + FMA work in the compiler-scheduled K loop cost more than the pass: 575 vs
// 636 TFLOP/s at Nt = 576 with the pass included; the factors therefore stay a pass of their own, computed once per token stack.)
constexpr int A_BM = 128, A_BN = 128, A_BK = 64;
constexpr int A_TILE = A_BM * A_BK * 2, A_STAGE = 2 * A_TILE, A_LDS = 2 * A_STAGE;

__global__ __launch_bounds__(256, 2) void ascore_maxcos_tiled(const AScoreArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntt = (p.Nt + A_BM - 1) / A_BM, nnt = (p.Nr + A_BN - 1) / A_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_img = ntt * nnt;
    const int img = bid / per_img, rem = bid - img * per_img;
    const int tt = rem / nnt, nt = rem - tt * nnt;
    const bf16_t* other = reinterpret_cast<const bf16_t*>(p.other) + (size_t)img * p.Nt * p.D;
    const bf16_t* ref = reinterpret_cast<const bf16_t*>(p.ref) + (size_t)img * p.Nr * p.D;
    const float* cr = p.c_ref + (size_t)img * p.Nr;
    const int m0 = tt * A_BM, n0 = nt * A_BN;
    const int srow = tid >> 3;
    const int lslot = (tid & 7) ^ ((srow >> 1) & 7);
    const bf16_t* ga[4];
    const bf16_t* gr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = m0 + j * 32 + srow;
        r = r < p.Nt ? r : p.Nt - 1;
        ga[j] = other + (size_t)r * p.D + lslot * 8;
        int q = n0 + j * 32 + srow;
        q = q < p.Nr ? q : p.Nr - 1;
        gr[j] = ref + (size_t)q * p.D + lslot * 8;
    }
    const int fr = lane & 15, fg = lane >> 4;
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = fr * 128 + (((kk * 4 + fg) ^ ((fr >> 1) & 7)) << 4);
    const int aoff = wm * 64 * 128, woff = A_TILE + wn * 64 * 128;
    float rowmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const int nk = p.D / A_BK;
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * A_STAGE + wave * 1024;
        char* sw = sa + A_TILE;
        const int ko = kt * A_BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(ga[j] + ko, sa + j * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(gr[j] + ko, sw + j * 4096);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // edge tiles: a wave whose 64 target rows or 64 reference columns lie entirely past the end (576 = 4.5 tiles: half of the last row /
    // column tile) skips its fragment reads and MFMAs; it still stages and meets the barriers
    const bool live = m0 + wm * 64 < p.Nt && n0 + wn * 64 < p.Nr;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sb = smem + cur * A_STAGE;
        if (live)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xa[4], xr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(sb + aoff + i * 2048 + foff[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(sb + woff + i * 2048 + foff[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xr[j], xa[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float* red = reinterpret_cast<float*>(smem);                 // the staging buffers are dead now (last __syncthreads above)
    // lane holds G[t = m0 + 64 wm + 16 i + fr][s = n0 + 64 wn + 16 j + 4 fg + e]: scale by c_ref[s], mask s >= Nr, row max
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sidx = n0 + wn * 64 + j * 16 + fg * 4;
        float c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = (sidx + e < p.Nr) ? cr[sidx + e] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (sidx + e < p.Nr) rowmax[i] = fmaxf(rowmax[i], acc[i][j][e] * c[e]);
    }
    // combine over the four 4-column groups (lanes fr + 16 fg), then over the two column waves; one float per (target row, ref tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = rowmax[i];
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        rowmax[i] = v;
    }
    __syncthreads();
    if (fg == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wn * 2 + wm) * 64 + i * 16 + fr] = rowmax[i];
    }
    __syncthreads();
    if (tid < 128) {
        const int t = m0 + tid;                                     // tid = 64 wm + 16 i + fr
        if (t < p.Nt) p.partial[((size_t)img * p.Nt + t) * nnt + nt] = fmaxf(red[(tid >> 6) * 64 + (tid & 63)], red[(2 + (tid >> 6)) * 64 + (tid & 63)]);
    }
}

// ---- persistent ping-pong Gram (round 3): the structure of the default GEMM (gemm_bf16_v5.hip) at tile sizes that divide the token
// stacks.  576 = 4.5 tiles of 128: the 128 x 128 kernel above launches 25 tile pairs per image for 20.25 tile pairs of work, and two
// 128 x 128 workgroups per CU pull 64 KB of operands per 1024 matrix-pipe cycles through the CU's L2 port (62 B/clk at MFMA peak).
// Here a tile is TM x TN with TM, TN in {192, 256} picked per operand (576 = 3 x 192, 256 = 1 x 256, 729 -> 768): no dead area on
// the encoder shapes of the sweep.  Eight waves in two groups of four skewed by one barrier (one wave per SIMD is always in an MFMA
// segment), wave tile (TM / 2) x (TN / 4) = MI x NJ accumulators of 16x16x32, K-tiles of 64 in 128-byte LDS rows, a ring of five 32-KB
// operand slots fed by LDS-DMA with counted waits (X0 W0 X1 W1 ...: while tile s is consumed W(s+1) and X(s+2) stream in), hand-written
// fragment reads (MI + NJ ds_read_b128 per k-half against MI * NJ MFMAs), persistent XCD-contiguous tile walk: the 32 workgroups of
// an XCD work on 32 consecutive tiles (a few images) and walk D in step, so an operand slab is fetched from HBM once.  The barrier /
// hazard ledger is gemm_bf16_v5.hip's (OWN_ = false) with TM / 64 and TN / 64 LDS-DMA pieces per wave instead of four.
// Epilogue: scale by c_ref, mask s >= Nr, row max over the wave's TN / 4 reference rows -> partial[img][t][4 nt + wn]; the finalize
// kernel takes the max over the 4 * nnt parts.
constexpr int P_TK = 64;

VR_DEV unsigned a_lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
// Fragment reads and their wait are asm: hipcc cannot tell an LDS read from the bytes an in-flight LDS-DMA will write and would put
// s_waitcnt vmcnt(0) in front of every read it can see, draining the prefetch ring.  The wait names every destination read-write, so no
// consumer (or copy) of a fragment can be scheduled above it; ordering against the DMA is the ledger of gemm_bf16_v5.hip.
template <int OFF> VR_DEV void a_read(bf16x8& f, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF)); }
template <int N> VR_DEV void a_reads(bf16x8 (&f)[N], unsigned addr) {
    a_read<0>(f[0], addr); a_read<2048>(f[1], addr); a_read<4096>(f[2], addr);
    if constexpr (N > 3) a_read<6144>(f[3], addr);
    if constexpr (N > 4) { a_read<8192>(f[4], addr); a_read<10240>(f[5], addr); }
    if constexpr (N > 6) { a_read<12288>(f[6], addr); a_read<14336>(f[7], addr); }
}
#define A_PIN3(f) "+v"(f[0]), "+v"(f[1]), "+v"(f[2])
#define A_PIN4(f) A_PIN3(f), "+v"(f[3])
#define A_PIN6(f) A_PIN4(f), "+v"(f[4]), "+v"(f[5])
#define A_PIN8(f) A_PIN6(f), "+v"(f[6]), "+v"(f[7])
template <int N> VR_DEV void a_pin(bf16x8 (&f)[N], bool wait) {      // wait: s_waitcnt lgkmcnt(0) in the statement; else a pure ordering point
    static_assert(N == 3 || N == 4 || N == 6 || N == 8, "fragment count");
    if (wait) {
        if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(0)" : A_PIN3(f));
        if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : A_PIN4(f));
        if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(0)" : A_PIN6(f));
        if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(0)" : A_PIN8(f));
    } else {
        if constexpr (N == 3) asm volatile("" : A_PIN3(f));
        if constexpr (N == 4) asm volatile("" : A_PIN4(f));
        if constexpr (N == 6) asm volatile("" : A_PIN6(f));
        if constexpr (N == 8) asm volatile("" : A_PIN8(f));
    }
}
VR_DEV void a_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> VR_DEV void a_wait_vm() {
    static_assert(N == 3 || N == 4 || N == 6, "pieces in flight");
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

template <int MI, int NJ, bool DEEP>
__global__ __launch_bounds__(512, 2) void ascore_maxcos_pp(const AScoreArgs p) {
    constexpr int TM = 32 * MI, TN = 64 * NJ;                  // 192 or 256 target rows x 192 or 256 reference rows per tile
    constexpr int PX = TM / 64, PW = TN / 64;                  // 8-row LDS-DMA pieces per wave and K-tile
    // DEEP (192 x 192 only: 24-KB slots): a ring of SIX slots, the stream runs two whole K-tiles ahead (prologue X0 W0 X1 W1; L0(s) issues
    // X(s+2), L1(s) issues W(s+2); the counted waits leave PX + PW pieces in flight).  X(s+2) / W(s+2) reuse the slots of X(s-1) / W(s-1), last
    // read in L1(s-1): retired before barrier instance 4s, overwritten after it - the ledger of gemm_bf16_v5.hip with one more item in flight.
    constexpr int SLOT = (TM > TN ? TM : TN) * P_TK * 2, NS = DEEP ? 6 : 5, INFL = DEEP ? PX + PW : PX;
    static_assert(SLOT * NS <= 160 * 1024, "LDS ring");
    static_assert((MI == 6 || MI == 8) && (NJ == 3 || NJ == 4), "tile shapes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int ntt = (p.Nt + TM - 1) / TM, nnt = (p.Nr + TN - 1) / TN, per_img = ntt * nnt, ntiles = p.n_img * per_img;
    // ---- persistent tile list: XCD-contiguous chunks (block b runs on XCD b % 8), strided by the blocks of that XCD
    int t_start, t_stride, t_count;
    {
        const int G = gridDim.x;
        const int nx = G < 8 ? G : 8;
        const int x = blockIdx.x % nx, j = blockIdx.x / nx;
        const int per = (G + nx - 1 - x) / nx;
        const int q = ntiles / nx, r = ntiles % nx;
        const int cstart = x * q + (x < r ? x : r), csize = q + (x < r ? 1 : 0);
        t_start = cstart + j;
        t_stride = per;
        t_count = j < csize ? (csize - j + per - 1) / per : 0;
    }
    if (t_count == 0) return;                                  // uniform per block: no barrier has been executed yet
    auto decode = [&](int i, int& img, int& m0, int& n0, int& nt) {
        const int ii = i < t_count ? i : t_count - 1;          // run-ahead loads past the end re-read the last tile (never consumed)
        const int t = t_start + ii * t_stride;
        img = t / per_img;
        const int rem = t - img * per_img;
        const int tt = rem / nnt;
        nt = rem - tt * nnt;
        m0 = tt * TM;
        n0 = nt * TN;
    };
    const int nk = p.D / P_TK;
    const int S = t_count * nk;

    // ---- LDS-DMA cursors: wave w stages rows [8 PX w, 8 PX (w + 1)) of the target tile and [8 PW w, ..) of the reference tile, 8 rows x
    //      128 B per instruction; lane -> (row = base + 8 j + lane>>3, physical slot lane&7) fetches logical slot (lane&7) ^ ((row>>1)&7)
    struct Cur { const bf16_t* q[4]; int k, ti, idx; };
    Cur cx, cw;
    const bf16_t* other = reinterpret_cast<const bf16_t*>(p.other);
    const bf16_t* refp = reinterpret_cast<const bf16_t*>(p.ref);
    auto set_x = [&](Cur& c) {
        int img, m0, n0, nt; decode(c.ti, img, m0, n0, nt);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int rt = wave * (8 * PX) + j * 8 + (lane >> 3);
            int r = m0 + rt; r = r < p.Nt ? r : p.Nt - 1;       // rows past the end are computed and masked
            c.q[j] = other + ((size_t)img * p.Nt + r) * p.D + (((lane & 7) ^ ((rt >> 1) & 7)) << 3);
        }
    };
    auto set_w = [&](Cur& c) {
        int img, m0, n0, nt; decode(c.ti, img, m0, n0, nt);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int rt = wave * (8 * PW) + j * 8 + (lane >> 3);
            int r = n0 + rt; r = r < p.Nr ? r : p.Nr - 1;
            c.q[j] = refp + ((size_t)img * p.Nr + r) * p.D + (((lane & 7) ^ ((rt >> 1) & 7)) << 3);
        }
    };
    cx.k = cw.k = 0; cx.ti = cw.ti = 0; cx.idx = cw.idx = 0;
    set_x(cx); set_w(cw);
    auto issue_x = [&]() {
        char* dst = smem + ((2 * cx.idx) % NS) * SLOT + wave * (1024 * PX);
#pragma unroll
        for (int j = 0; j < PX; ++j) glds16(cx.q[j] + cx.k, dst + j * 1024);
        ++cx.idx; cx.k += P_TK;
        if (cx.k == p.D) { cx.k = 0; ++cx.ti; set_x(cx); }
    };
    auto issue_w = [&]() {
        char* dst = smem + ((2 * cw.idx + 1) % NS) * SLOT + wave * (1024 * PW);
#pragma unroll
        for (int j = 0; j < PW; ++j) glds16(cw.q[j] + cw.k, dst + j * 1024);
        ++cw.idx; cw.k += P_TK;
        if (cw.k == p.D) { cw.k = 0; ++cw.ti; set_w(cw); }
    };
    // ---- fragment reads: row = base16 + (lane&15), logical slot 4 h + (lane>>4), physical = logical ^ ((row>>1)&7); 16-row steps (and the
    //      group / wave bases, multiples of 16 rows) leave (row>>1)&7 alone, the k-half only flips slot bit 2
    const int fr = lane & 15, hi = lane >> 4;
    const int fbase = fr * 128 + ((hi ^ ((fr >> 1) & 7)) << 4);
    const int xbase = grp * (16 * MI) * 128 + fbase;
    const int wbase = wn * (16 * NJ) * 128 + fbase;

    // ---- prologue: X0, W0 landed and visible, X1 in flight
    issue_x(); issue_w(); issue_x();
    if (DEEP) issue_w();
    a_wait_vm<INFL>();
    a_barrier();
    const unsigned lds0 = a_lds_addr(smem);
    auto body = [&](auto G_) __attribute__((always_inline)) {
        constexpr int G = decltype(G_)::value;
        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int kt = 0, ti = 0;
        if (G == 1) a_barrier();                               // skew: group 1 runs one barrier interval behind
        for (int s = 0; s < S; ++s) {
            const unsigned sx = lds0 + (unsigned)((2 * s) % NS) * SLOT, sw = lds0 + (unsigned)((2 * s + 1) % NS) * SLOT;
            bf16x8 xf[MI], wf[NJ];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // L(h): the k-half's fragment reads + the LDS-DMA pieces of W(s+1) (h = 0) / X(s+2) (h = 1)
                __builtin_amdgcn_s_setprio(1);
                a_reads(wf, (sw + wbase) ^ (h << 6));           // W first: the MFMA segment starts with wf[0..] x xf[0]
                a_reads(xf, (sx + xbase) ^ (h << 6));
                if ((h == 0) != DEEP) issue_w(); else issue_x();
                if (h == 1 && G == 1) a_wait_vm<INFL>();        // group 1's loads are needed by group 0 one barrier later
                a_pin(xf, true);
                a_pin(wf, false);
                __builtin_amdgcn_s_setprio(0);
                a_barrier();
                // M(h): MI x NJ MFMAs and nothing else
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                if (h == 0) a_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);                  // keep the counted wait behind the segment's MFMAs
            if (G == 0) a_wait_vm<INFL>();
            if (++kt == nk) {
                kt = 0;
                int img, m0, n0, nt; decode(ti, img, m0, n0, nt); ++ti;
                const int mb = m0 + grp * (16 * MI), nb = n0 + wn * (16 * NJ);
                const float* cr = p.c_ref + (size_t)img * p.Nr;
                float rm[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) rm[i] = -INFINITY;
                // lane holds G[t = mb + 16 i + fr][s = nb + 16 j + 4 hi + e]
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int sidx = nb + j * 16 + hi * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = sidx + e < p.Nr;
                        const float c = ok ? cr[sidx + e] : 0.f;
#pragma unroll
                        for (int i = 0; i < MI; ++i) rm[i] = fmaxf(rm[i], ok ? acc[i][j][e] * c : -INFINITY);
                    }
                }
                const int nparts = 4 * nnt;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    float v = rm[i];
                    v = fmaxf(v, __shfl_xor(v, 16));
                    v = fmaxf(v, __shfl_xor(v, 32));
                    const int t = mb + i * 16 + fr;
                    if (hi == 0 && t < p.Nt) p.partial[((size_t)img * p.Nt + t) * nparts + nt * 4 + wn] = v;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            a_barrier();
        }
        if (G == 0) a_barrier();                               // group 1 executed one extra barrier up front
    };
    if (grp == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // drain the (unused) run-ahead loads before exit
}

template <int MI, int NJ, bool DEEP>
int launch_pp(const AScoreArgs& a, float* scores, hipStream_t s) {
    constexpr int TM = 32 * MI, TN = 64 * NJ, P_LDS = (TM > TN ? TM : TN) * P_TK * 2 * (DEEP ? 6 : 5);
    static VisrepLdsOptIn opt;                                   // per (kernel instantiation, device)
    visrep_lds_opt_in(opt, reinterpret_cast<const void*>(ascore_maxcos_pp<MI, NJ, DEEP>), P_LDS);
    const int ncu = visrep_cu_count();
    const int nnt = (a.Nr + TN - 1) / TN, ntiles = a.n_img * ((a.Nt + TM - 1) / TM) * nnt;
    hipLaunchKernelGGL((ascore_maxcos_pp<MI, NJ, DEEP>), dim3(ntiles < ncu ? ntiles : ncu), dim3(512), P_LDS, s, a);
    return 4 * nnt;                                            // parts per target row in a.partial
}

// score[img] = (1 / Nt) * sum_t c_other[t] * max over the reference tiles of rowmax[img][t][.]   (c_other > 0 commutes with the max);
// one wave per image, lanes stride t, fixed-order wave reduction -> deterministic
__global__ __launch_bounds__(256) void ascore_finalize_tiles(const float* __restrict__ rowmax, const float* __restrict__ c_other, float* __restrict__ score,
                                                             int n_img, int Nt, int nnt) {
    const int lane = threadIdx.x & 63;
    const int img = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (img >= n_img) return;
    float s = 0.f;
    for (int t = lane; t < Nt; t += 64) {
        const float* r = rowmax + ((size_t)img * Nt + t) * nnt;
        float m = r[0];
        for (int k = 1; k < nnt; ++k) m = fmaxf(m, r[k]);
        s += m * c_other[(size_t)img * Nt + t];
    }
    s = wave_sum(s);
    if (lane == 0) score[img] = s / (float)Nt;
}

template <typename T>
int row_scales(const void* x, long rows, int D, float* scale, hipStream_t s) {
    hipLaunchKernelGGL(ascore_row_scale<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const T*)x, rows, D, scale);
    return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
}

// c_other / c_ref: row scales computed earlier (visrep_ascore_row_scale) or null -> computed here into the workspace
template <typename T>
int run(const void* other, const void* ref, const float* c_other_in, const float* c_ref_in, int n_img, int Nt, int Nr, int D, float* scores,
        float* ws, hipStream_t s) {
    float* c_other = ws;
    float* c_ref = c_other + (size_t)n_img * Nt;
    float* partial = c_ref + (size_t)n_img * Nr;
    const bool tiled = sizeof(T) == 2 && D % 64 == 0;
    if (!c_other_in && row_scales<T>(other, (long)n_img * Nt, D, c_other, s)) return VISREP_ERR_LAUNCH;
    if (!c_ref_in && row_scales<T>(ref, (long)n_img * Nr, D, c_ref, s)) return VISREP_ERR_LAUNCH;
    AScoreArgs a{other, ref, c_other_in ? c_other_in : c_other, c_ref_in ? c_ref_in : c_ref, partial, n_img, Nt, Nr, D};
    int ntt = (Nt + 63) / 64;
    // tile shape: the candidate that launches the smallest tile area; ties go to the ping-pong kernel and there to the larger tile
    auto cover = [](int n, int t) { return (long)((n + t - 1) / t) * t; };
    const int tm = cover(Nt, 256) <= cover(Nt, 192) ? 256 : 192, tn = cover(Nr, 256) <= cover(Nr, 192) ? 256 : 192;
    const long area128 = cover(Nt, 128) * cover(Nr, 128), area_pp = cover(Nt, tm) * cover(Nr, tn);
    const bool pp = t_visrep_ascore_variant == 2 || (t_visrep_ascore_variant == 0 && area_pp <= area128);
    if (tiled && pp) {
        const int parts = tm == 192 ? (tn == 192 ? launch_pp<6, 3, ASCORE_DEEP != 0>(a, scores, s)
                                                 : launch_pp<6, 4, false>(a, scores, s))
                                    : (tn == 192 ? launch_pp<8, 3, false>(a, scores, s) : launch_pp<8, 4, false>(a, scores, s));
        hipLaunchKernelGGL(ascore_finalize_tiles, dim3((n_img + 3) / 4), dim3(256), 0, s, partial, a.c_other, scores, n_img, Nt, parts);
        return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
    }
    if (tiled) {                                               // LDS-tiled MFMA kernel, one 128 x 128 tile per workgroup
        ntt = (Nt + A_BM - 1) / A_BM;
        const int nnt = (Nr + A_BN - 1) / A_BN;
        static VisrepLdsOptIn opt;
        visrep_lds_opt_in(opt, reinterpret_cast<const void*>(ascore_maxcos_tiled), A_LDS);
        hipLaunchKernelGGL(ascore_maxcos_tiled, dim3((unsigned)((size_t)n_img * ntt * nnt)), dim3(256), A_LDS, s, a);
        hipLaunchKernelGGL(ascore_finalize_tiles, dim3((n_img + 3) / 4), dim3(256), 0, s, partial, a.c_other, scores, n_img, Nt, nnt);
        return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(ascore_maxcos<T>, dim3(n_img * ntt), dim3(256), 0, s, a);
    hipLaunchKernelGGL(ascore_finalize, dim3((n_img + 63) / 64), dim3(64), 0, s, partial, scores, n_img, ntt, Nt);
    return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
}

}  // namespace

extern "C" size_t visrep_ascore_workspace_bytes(int n_img, int Nt, int Nr) {
    // row scales of both operands + the larger of the two partial layouts: [n_img, ceil(Nt / 64)] block sums (direct kernels) or
    // [n_img, Nt, parts] per-row maxima of the tiled bf16 kernels (one part per 128-row reference tile, or four per 192- / 256-row tile)
    const size_t parts = (size_t)4 * ((Nr + 191) / 192);       // >= ceil(Nr / 128) and >= 4 * ceil(Nr / 256): covers every tile shape
    const size_t direct = (size_t)n_img * ((Nt + 63) / 64), tiled = (size_t)n_img * Nt * parts;
    return sizeof(float) * ((size_t)n_img * Nt + (size_t)n_img * Nr + (direct > tiled ? direct : tiled));
}

extern "C" int visrep_set_ascore_variant(int v) {
    if (v < 0 || v > 2) return visrep_set_error(VISREP_ERR_SHAPE, "ascore variant: 0 (by tile area), 1 (128 x 128 tiles) or 2 (persistent ping-pong tiles)");
    const int old = t_visrep_ascore_variant;                   // per-thread state
    t_visrep_ascore_variant = v;
    return old;
}

extern "C" int visrep_ascore_row_scale(const void* x, long rows, int D, int dtype, float* scale, void* stream) {
    if (rows <= 0) return 0;
    if (!x || !scale) return visrep_set_error(VISREP_ERR_ARG, "ascore_row_scale: null pointer");
    if (dtype == VISREP_BF16 && D > 0 && D % 16 == 0) {
        return row_scales<bf16_t>(x, rows, D, scale, (hipStream_t)stream) ? visrep_set_error(VISREP_ERR_LAUNCH, "ascore_row_scale: launch failed") : 0;
    }
    if (dtype == VISREP_F32 && D > 0 && D % 8 == 0) {
        return row_scales<float>(x, rows, D, scale, (hipStream_t)stream) ? visrep_set_error(VISREP_ERR_LAUNCH, "ascore_row_scale: launch failed") : 0;
    }
    return visrep_set_error(VISREP_ERR_SHAPE, "ascore_row_scale: VISREP_BF16 with D % 16 == 0 or VISREP_F32 with D % 8 == 0");
}

extern "C" int visrep_ascore_maxcos_scaled(const void* other, const void* ref, const float* other_scale, const float* ref_scale, int n_img,
                                           int Nt, int Nr, int D, int dtype, float* scores, void* workspace, void* stream) {
    if (n_img <= 0) return 0;
    if (Nt <= 0 || Nr <= 0 || D <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "ascore: empty tensor");
    if (dtype == VISREP_BF16) {
        if (D % 16) return visrep_set_error(VISREP_ERR_SHAPE, "ascore: bf16 path needs D % 16 == 0");
        const int rc = run<bf16_t>(other, ref, other_scale, ref_scale, n_img, Nt, Nr, D, scores, (float*)workspace, (hipStream_t)stream);
        return rc ? visrep_set_error(rc, "ascore: launch failed") : 0;
    }
    if (dtype == VISREP_F32) {
        if (D % 8) return visrep_set_error(VISREP_ERR_SHAPE, "ascore: fp32 path needs D % 8 == 0");
        const int rc = run<float>(other, ref, other_scale, ref_scale, n_img, Nt, Nr, D, scores, (float*)workspace, (hipStream_t)stream);
        return rc ? visrep_set_error(rc, "ascore: launch failed") : 0;
    }
    return visrep_set_error(VISREP_ERR_ARG, "ascore: dtype must be VISREP_BF16 or VISREP_F32");
}

extern "C" int visrep_ascore_maxcos(const void* other, const void* ref, int n_img, int Nt, int Nr, int D, int dtype, float* scores,
                                    void* workspace, void* stream) {
    return visrep_ascore_maxcos_scaled(other, ref, nullptr, nullptr, n_img, Nt, Nr, D, dtype, scores, workspace, stream);
}
