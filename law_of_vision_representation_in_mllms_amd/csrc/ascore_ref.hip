// A score in the REFERENCE'S OWN ARITHMETIC on bf16 tensors (gfx950).
//
// /root/reference/A_score/compute.py runs on the tensors as they were dumped - bf16 (LLaVA's model.to(bfloat16)) - and torch
// evaluates every op of compute.py:12-15,54-72 in that dtype: each op computes in fp32 and ROUNDS ITS RESULT TO bf16.  That is why the
// published table (policy/ablations_t.csv) holds values like 1.0078125 = 1 + 2^-7 for CLIP336 against itself.  The default A-score kernels
// (ascore.hip) compute the same quantity with exact products and fp32 accumulation - closer to the real cosine, but a third-digit difference
// from what the reference PRINTS.  This file reproduces the printed numbers: op by op,
//
//   normalize_feat (compute.py:12-15)      n  = bf16(sqrt(sum_d x^2));  n' = bf16(n + 1e-10);  o = bf16(x / n')
//   F.cosine_similarity (compute.py:64-65; ATen CosineSimilarity: x / max(|x|, eps) per operand, product, sum over dim)
//                                          m  = max(bf16(sqrt(sum_d o^2)), bf16(1e-8));  u = bf16(o / m)
//                                          S[t, s] = bf16( sum_d bf16(u_other[t, d] * u_ref[s, d]) )       (fp32 running sum of ROUNDED products)
//   .max(dim=1).values.mean() (compute.py:68-72; ATen mean of a bf16 tensor = fp32 sum -> fp32 divide -> one rounding)
//                                          score = bf16( (sum_t max_s S[t, s]) / Nt )
//
// The products are rounded one by one, so this is NOT a matrix product: it runs on the VALU (v_mul_f32 - exact for two bf16 factors -,
// v_cvt_pk_bf16_f32, fp32 adds), ~3.5 instructions per element pair; 576 x 576 x 4096 per image is ~20 us of chip time.  A parity mode
// (ascore_ops.max_cos_mean(..., arithmetic="reference")), not the throughput path.  What can differ from torch's CPU result is only the
// ORDER of the fp32 sums (1e-7 relative), visible when a sum lands within that distance of a bf16 rounding boundary (~3e-5 of the entries).
#include "common.h"
#include "visrep_internal.h"

namespace {

VR_DEV float bf16r(float v) { return bf2f(f2bf(v)); }           // round to nearest even bf16, back in fp32

// one wave per row: both normalisations of a token row, rounded like torch's bf16 ops; y may not alias x
__global__ __launch_bounds__(256) void ascore_ref_rows(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + row * D;
    bf16_t* yr = y + row * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = bf2f(xr[d]); ss = __builtin_fmaf(v, v, ss); }
    ss = wave_sum(ss);
    float n = bf16r(__fsqrt_rn(ss));
    n = bf16r(n + 1e-10f);                                       // compute.py:14-15: norms + epsilon (a bf16 tensor + python scalar)
    float ss2 = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float o = bf16r(__fdiv_rn(bf2f(xr[d]), n));        // feat / (norms + epsilon)
        yr[d] = f2bf(o);
        ss2 = __builtin_fmaf(o, o, ss2);
    }
    ss2 = wave_sum(ss2);
    const float m = fmaxf(bf16r(__fsqrt_rn(ss2)), bf16r(1e-8f)); // cosine_similarity: x_norm.clamp_min(eps) in the tensor's dtype
    for (int d = lane; d < D; d += 64) yr[d] = f2bf(__fdiv_rn(bf2f(yr[d]), m));       // each lane re-reads only what it wrote itself
}

constexpr int RT = 64, RK = 64, RLD = RK + 8;                    // 64 x 64 score tile, 64-deep chunks, LDS rows padded by 16 B

// S tile of one image: thread (ty, tx) owns rows 4 ty .. + 3 and columns 4 tx .. + 3; rowmax[img][t][tile_s] = max over the tile's columns of bf16(S)
__global__ __launch_bounds__(256) void ascore_ref_gram(const bf16_t* __restrict__ u_other, const bf16_t* __restrict__ u_ref, float* __restrict__ rowmax,
                                                       int Nt, int Nr, int D, int ntt, int nst) {
    __shared__ __attribute__((aligned(16))) bf16_t sa[RT][RLD];
    __shared__ __attribute__((aligned(16))) bf16_t sb[RT][RLD];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int b = blockIdx.x;
    const int img = b / (ntt * nst), r = b - img * ntt * nst, t0 = (r / nst) * RT, s0 = (r % nst) * RT, ts = r % nst;
    const bf16_t* A = u_other + (size_t)img * Nt * D;
    const bf16_t* B = u_ref + (size_t)img * Nr * D;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int lr = tid >> 2, lc = (tid & 3) * 16;                // staging: thread -> (row, 16 elements)
    for (int k0 = 0; k0 < D; k0 += RK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int d = k0 + lc + e;
            sa[lr][lc + e] = (t0 + lr < Nt && d < D) ? A[(size_t)(t0 + lr) * D + d] : (bf16_t)0;
            sb[lr][lc + e] = (s0 + lr < Nr && d < D) ? B[(size_t)(s0 + lr) * D + d] : (bf16_t)0;
        }
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < RK; kk += 8) {
            u32x4 av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const u32x4*>(&sa[ty * 4 + i][kk]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const u32x4*>(&sb[tx * 4 + j][kk]);
#pragma unroll
            for (int w = 0; w < 4; ++w)                           // two elements per dword, ascending d
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t p = pack_bf16(bf_lo(av[i][w]) * bf_lo(bv[j][w]), bf_hi(av[i][w]) * bf_hi(bv[j][w]));   // products exact in fp32, rounded to bf16
                        acc[i][j] += bf_lo(p);
                        acc[i][j] += bf_hi(p);
                    }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (s0 + tx * 4 + j < Nr) m = fmaxf(m, bf16r(acc[i][j]));      // S is a bf16 tensor: the max sees rounded values
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));    // the 16 lanes of a row group
        const int t = t0 + ty * 4 + i;
        if (tx == 0 && t < Nt) rowmax[((size_t)img * Nt + t) * nst + ts] = m;
    }
}

// score[img] = bf16( (sum_t max over tiles) / Nt ): one workgroup per image, fixed summation order
__global__ __launch_bounds__(256) void ascore_ref_finalize(const float* __restrict__ rowmax, float* __restrict__ score, int Nt, int nst) {
    __shared__ float part[4];
    const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    for (int t = threadIdx.x; t < Nt; t += 256) {
        const float* r = rowmax + ((size_t)img * Nt + t) * nst;
        float m = r[0];
        for (int k = 1; k < nst; ++k) m = fmaxf(m, r[k]);
        s += m;
    }
    s = wave_sum(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) score[img] = bf16r(__fdiv_rn(((part[0] + part[1]) + part[2]) + part[3], (float)Nt));
}

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" size_t visrep_ascore_refarith_workspace_bytes(int n_img, int Nt, int Nr, int D) {
    if (n_img <= 0 || Nt <= 0 || Nr <= 0 || D <= 0) return 0;
    return up256((size_t)n_img * Nt * D * 2) + up256((size_t)n_img * Nr * D * 2) + up256((size_t)n_img * Nt * ((Nr + RT - 1) / RT) * sizeof(float));
}

extern "C" int visrep_ascore_maxcos_refarith(const void* other, const void* ref, int n_img, int Nt, int Nr, int D, float* scores, void* workspace,
                                             void* stream) {
    if (n_img <= 0) return 0;
    if (!other || !ref || !scores || !workspace) return visrep_set_error(VISREP_ERR_ARG, "ascore_refarith: null pointer");
    if (Nt <= 0 || Nr <= 0 || D <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "ascore_refarith: empty tensor");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    bf16_t* uo = (bf16_t*)ws;
    bf16_t* ur = (bf16_t*)(ws + up256((size_t)n_img * Nt * D * 2));
    float* rowmax = (float*)(ws + up256((size_t)n_img * Nt * D * 2) + up256((size_t)n_img * Nr * D * 2));
    const int ntt = (Nt + RT - 1) / RT, nst = (Nr + RT - 1) / RT;
    const long ro = (long)n_img * Nt, rr = (long)n_img * Nr;
    hipLaunchKernelGGL(ascore_ref_rows, dim3((unsigned)((ro + 3) / 4)), dim3(256), 0, s, (const bf16_t*)other, uo, ro, D);
    hipLaunchKernelGGL(ascore_ref_rows, dim3((unsigned)((rr + 3) / 4)), dim3(256), 0, s, (const bf16_t*)ref, ur, rr, D);
    hipLaunchKernelGGL(ascore_ref_gram, dim3((unsigned)((size_t)n_img * ntt * nst)), dim3(256), 0, s, uo, ur, rowmax, Nt, Nr, D, ntt, nst);
    hipLaunchKernelGGL(ascore_ref_finalize, dim3(n_img), dim3(256), 0, s, rowmax, scores, Nt, nst);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "ascore_refarith: launch failed");
}
