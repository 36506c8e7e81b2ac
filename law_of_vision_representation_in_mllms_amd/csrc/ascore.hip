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
#include "common.h"
#include "visrep_internal.h"

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
    if (tiled) {                                               // production path: LDS-tiled MFMA kernel, one 128 x 128 tile per workgroup
        ntt = (Nt + A_BM - 1) / A_BM;
        const int nnt = (Nr + A_BN - 1) / A_BN;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ascore_maxcos_tiled), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS); attr = true; }
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
    // [n_img, Nt, ceil(Nr / 128)] per-row maxima of the tiled bf16 kernel
    const size_t direct = (size_t)n_img * ((Nt + 63) / 64), tiled = (size_t)n_img * Nt * ((Nr + 127) / 128);
    return sizeof(float) * ((size_t)n_img * Nt + (size_t)n_img * Nr + (direct > tiled ? direct : tiled));
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
