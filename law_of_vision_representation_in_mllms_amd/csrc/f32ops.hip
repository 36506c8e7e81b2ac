// Reference-precision (fp32) tower path for gfx950.
//
// Why it exists: the reference builds the C-score CLIP / OpenCLIP / DINOv2 towers WITHOUT a dtype cast and feeds them fp32
// pixels (C_score/extract_feature.py:36-45,49-50,80-87) - only SigLIP and the diffusion towers run in bf16 there.  The bf16
// MFMA engine perturbs features by ~1e-2 relative, which moves the A score by ~1e-3 and can flip PCK hits; this path keeps every
// tensor and every accumulation in fp32 so that images -> tower -> scores can be compared with the fp32 reference chain at 1e-4.
//
// Two routes (engine.VitEngineF32 picks; DESIGN.md section 6, profiles/round3_f32.md):
//   exact  visrep_vit_forward_f32: v_mfma_f32_32x32x2_f32 - fp32 operands, fp32 accumulate, bitwise a chain of fmaf (MI355X_MICROARCH.md,
//          "Matrix cores"; 157 TFLOP/s peak, 1/16 of the bf16 rate).  One LDS-staged kernel (128x128x16 tiles, 4 waves, register prefetch of
//          the next K-step, asm-pipelined fragment reads) serves every contraction - patch embedding, QKV / out / MLP projections with fused
//          bias / activation / LayerScale + residual, batched / gathered products for the C-score post-processors - and a flash-style
//          attention keeps the fp32 scores in registers (head width 64; other widths: batched Q K^T -> softmax rows -> P V).  Any shape.
//   split  visrep_vit_forward_f32_split: the projections and the attention on the bf16 matrix pipe at fp32 accuracy - an fp32 value as
//          three bf16 planes (24 significand bits), six plane-pair products, fp32 accumulation and epilogues (gemm_bf16_v5.hip EPI_F32X,
//          attn_f32_split_kernel below): ~1.6x the exact route end to end, the same error level against float64.  d, mlp % 256 == 0 and
//          head width 64, i.e. every CLIP / OpenCLIP / DINOv2 tower of the paper.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "visrep_internal.h"

namespace {

struct GemmF32Args {
    const float* A; const float* W; float* C;
    const float* bias; const float* resid; const float* ls;
    int M, N, K, lda, ldw, ldc;
    int w_kn;            // 0: W is [N, K] (nn.Linear layout); 1: W is [K, N] (plain matrix product A B)
    int epi, act;        // EPI_BIAS / EPI_ACT / EPI_RESID
    float alpha;         // C = epi(alpha * A W + bias)
    int nb2;             // batches: blockIdx.z = b1 * nb2 + b2, element strides below
    long sA1, sA2, sW1, sW2, sC1, sC2;
    const int* idxA; const int* idxW;   // gather form (nb2 == 1): batch z reads A + idxA[z] * sA1 and W + idxW[z] * sW1
};

constexpr int FBM = 128, FBN = 128, FBK = 16, FLD = 132;   // FLD: LDS row stride (floats); 132 * 4 B keeps 16-B alignment, 2-way write conflicts at most

// LDS fragment pair (rows r and r + 32 of one k) and its counted wait as asm (see the K loop)
VR_DEV unsigned lds_addr_f32(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
VR_DEV void lds_read2_f32(f32x2& f, unsigned addr) { asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(f) : "v"(addr)); }
template <int N> VR_DEV void lds_wait2_f32(f32x2& a, f32x2& b) {
    if (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
    else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b));
}

VR_DEV float act_f32(float x, int act) {
    switch (act) {
        case ACT_QUICK_GELU: return x * (1.0f / (1.0f + expf(-1.702f * x)));          // x * sigmoid(1.702 x) (HF QuickGELUActivation)
        case ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_GELU_TANH: return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
        default: return x;
    }
}

// 16 consecutive k of one operand row -> 4 registers (float4) per chunk; rows / columns past the edge are clamped by the caller,
// the K tail is zero-filled element by element (never reads past a row's K elements)
VR_DEV float4 load_k4(const float* row, int k, int K) {
    if (k + 3 < K) return *reinterpret_cast<const float4*>(row + k);
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < K) v.x = row[k];
    if (k + 1 < K) v.y = row[k + 1];
    if (k + 2 < K) v.z = row[k + 2];
    return v;
}
VR_DEV float4 load_n4(const float* row, int n, int N, bool kvalid) {
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (!kvalid) return v;
    if (n + 3 < N) return *reinterpret_cast<const float4*>(row + n);
    if (n < N) v.x = row[n];
    if (n + 1 < N) v.y = row[n + 1];
    if (n + 2 < N) v.z = row[n + 2];
    return v;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Args p) {
    __shared__ __attribute__((aligned(16))) float As[2][FBK][FLD];
    __shared__ __attribute__((aligned(16))) float Ws[2][FBK][FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z - b1 * p.nb2;
    const float* A = p.A + (p.idxA ? (long)p.idxA[b1] : (long)b1) * p.sA1 + b2 * p.sA2;
    const float* W = p.W + (p.idxW ? (long)p.idxW[b1] : (long)b1) * p.sW1 + b2 * p.sW2;
    float* C = p.C + b1 * p.sC1 + b2 * p.sC2;
    const float* R = p.resid ? p.resid + b1 * p.sC1 + b2 * p.sC2 : nullptr;
    const int m0 = blockIdx.y * FBM, n0 = blockIdx.x * FBN;

    // staging map, k-contiguous operands: thread -> (row r + 64 j, k chunk c); 4 lanes cover one row's 64 B
    const int sr = tid >> 2, sc = tid & 3;
    // staging map, [K, N] operand: thread -> (k row kr + 8 j, 4 columns at 4 * nc)
    const int kr = tid >> 5, nc = tid & 31;
    const float* arow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + sr + 64 * j;
        arow[j] = A + (size_t)(m < p.M ? m : p.M - 1) * p.lda;
    }
    const float* wrow[2] = {nullptr, nullptr};
    if (!p.w_kn) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + sr + 64 * j;
            wrow[j] = W + (size_t)(n < p.N ? n : p.N - 1) * p.ldw;
        }
    }
    float4 ra[2], rw[2];
    // Interior K-steps (and, for the [K, N] operand, interior column blocks) take one branch-free path: the per-thread tail tests of
    // load_k4 / load_n4 inside the K loop cost a divergent branch + s_waitcnt per load (gemm_f32 ran at 0.61-0.68 of the fp32 MFMA roof
    // with them, profiles/round3_f32.md); only the last partial K-step / the last column block keeps the element-wise path.
    const bool n_interior = n0 + FBN <= p.N;                      // uniform
    auto fetch = [&](int k0) {
        const bool full = k0 + FBK <= p.K;                       // uniform
        if (full) {
#pragma unroll
            for (int j = 0; j < 2; ++j) ra[j] = *reinterpret_cast<const float4*>(arow[j] + k0 + 4 * sc);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) ra[j] = load_k4(arow[j], k0 + 4 * sc, p.K);
        }
        if (!p.w_kn) {
            if (full) {
#pragma unroll
                for (int j = 0; j < 2; ++j) rw[j] = *reinterpret_cast<const float4*>(wrow[j] + k0 + 4 * sc);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) rw[j] = load_k4(wrow[j], k0 + 4 * sc, p.K);
            }
        } else if (full && n_interior) {
#pragma unroll
            for (int j = 0; j < 2; ++j) rw[j] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + kr + 8 * j) * p.ldw + n0 + 4 * nc);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + kr + 8 * j;
                rw[j] = load_n4(W + (size_t)(k < p.K ? k : 0) * p.ldw, n0 + 4 * nc, p.N, k < p.K);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = sr + 64 * j;
            As[buf][4 * sc + 0][r] = ra[j].x; As[buf][4 * sc + 1][r] = ra[j].y;
            As[buf][4 * sc + 2][r] = ra[j].z; As[buf][4 * sc + 3][r] = ra[j].w;
        }
        if (!p.w_kn) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = sr + 64 * j;
                Ws[buf][4 * sc + 0][r] = rw[j].x; Ws[buf][4 * sc + 1][r] = rw[j].y;
                Ws[buf][4 * sc + 2][r] = rw[j].z; Ws[buf][4 * sc + 3][r] = rw[j].w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(&Ws[buf][kr + 8 * j][4 * nc]) = rw[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{};

    const int nk = (p.K + FBK - 1) / FBK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * FBK);                 // next K-step in flight under this step's MFMAs
        // one MFMA k-step = 2 k: lane half `hi` supplies k = 2 kk + hi.  The fragments of step kk + 1 are read BEFORE the four MFMAs of
        // step kk are issued.  hipcc serialises them whatever the source order (ds_read, lgkmcnt(0), 4 MFMAs, ds_read, ... - the MFMAs of a
        // step waited for that step's LDS latency), so the reads and their counted waits are asm: the compiler cannot re-merge them.
        const unsigned aaddr = lds_addr_f32(&As[buf][hi][wm * 64 + lq]), waddr = lds_addr_f32(&Ws[buf][hi][wn * 64 + lq]);
        f32x2 fa[2], fw[2];
        lds_read2_f32(fa[0], aaddr);
        lds_read2_f32(fw[0], waddr);
#pragma unroll
        for (int kk = 0; kk < FBK / 2; ++kk) {
            const int c = kk & 1;
            if (kk + 1 < FBK / 2) {
                lds_read2_f32(fa[c ^ 1], aaddr + (kk + 1) * 2 * FLD * 4);
                lds_read2_f32(fw[c ^ 1], waddr + (kk + 1) * 2 * FLD * 4);
                lds_wait2_f32<2>(fa[c], fw[c]);
            } else {
                lds_wait2_f32<0>(fa[c], fw[c]);
            }
            const float a0 = fa[c][0], a1 = fa[c][1], w0 = fw[c][0], w1 = fw[c][1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) stash(buf ^ 1);                        // the other buffer was last read before the previous barrier
        __syncthreads();
    }

    // epilogue: lane holds column n = .. + lq and rows (r & 3) + 8 (r >> 2) + 4 hi of each 32x32 accumulator
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + 32 * j + lq;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        const float lv = p.ls ? p.ls[n] : 1.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= p.M) continue;
                float v = p.alpha * acc[i][j][r] + bv;
                if (p.epi == EPI_ACT) v = act_f32(v, p.act);
                if (p.epi == EPI_RESID) v = R[(size_t)m * p.ldc + n] + lv * v;
                C[(size_t)m * p.ldc + n] = v;
            }
        }
    }
}

// Fused fp32 attention forward for head width 64 (flash style, exact-fp32 MFMA): replaces the three batched launches
// scores = scale Q K^T -> softmax rows -> P V with their [B, H, T, T] fp32 round trips through HBM (5.4 GB written, re-read three times per
// layer at batch 256: 27 % of the fp32 tower's time for 9 % of its FLOP, profiles/round3_f32.md) by one kernel whose scores never leave
// the registers.  Same structure as the bf16 attn_fwd<1> (attention.hip), fp32 operands:
//   * block = 4 waves x 32 query rows; key tiles of 64 start at the image's first token (fp32 Q | K | V are plain row-major [M, 3 d]
//     buffers, no transposed-V alignment constraint): only an image's LAST tile is masked;
//   * S^T = K Q^T with v_mfma_f32_32x32x2_f32 (A = K fragment from LDS, B = Q fragment: 32 registers hold this lane's Q row for the
//     whole kernel); lane (q = lane & 31, hi) then holds keys (r & 3) + 8 (r >> 2) + 4 hi of each 32-key block in register r;
//   * O^T += V^T P^T feeds register r of S as the B operand of k-step r: the MFMA's k index is a summation index, so both operands only
//     have to agree on the key - the A fragment is V[key(r, hi)][d] read straight from the row-major V tile;
//   * K / V tiles are [64][65] fp32 in LDS (the odd row stride makes the column reads of the K fragments conflict free), double
//     buffered, staged through registers; running maximum / sum in fp32, exp2 of the scaled difference like the bf16 kernel.
// Arithmetic is fp32 throughout; against HF's eager softmax the results differ by summation order only (~1e-6 relative).
}  // namespace
// diagnostics, per-thread (no process-global mutable state): bit 0 = the three-launch attention (batched Q K^T -> softmax rows -> P V) for
// every head width on the exact route; bit 1 = exact-fp32 MFMA attention inside the split route (visrep_debug_f32_attention)
thread_local int t_visrep_f32_attention_dbg = 0;
namespace {

// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): the subtractions are exact in fp32, so the three planes
// carry 24 significand bits - the operands of the split-bf16 GEMM (gemm_bf16_v5.hip EPI_F32X)
VR_DEV void split3(const float (&v)[4], u32x2& hi, u32x2& mid, u32x2& lo) {
    hi = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    const float r0 = v[0] - bf_lo(hi[0]), r1 = v[1] - bf_hi(hi[0]), r2 = v[2] - bf_lo(hi[1]), r3 = v[3] - bf_hi(hi[1]);
    mid = u32x2{pack_bf16(r0, r1), pack_bf16(r2, r3)};
    lo = u32x2{pack_bf16(r0 - bf_lo(mid[0]), r1 - bf_hi(mid[0])), pack_bf16(r2 - bf_lo(mid[1]), r3 - bf_hi(mid[1]))};
}

// fp32 [rows, K] (leading dimension ldx) -> bf16 planes [rows, NPL K] = hi | mid [| lo]; K % 4 == 0
template <int NPL>
__global__ __launch_bounds__(256) void split_bf16_planes_kernel(const float* __restrict__ x, int ldx, long rows, int K, bf16_t* __restrict__ planes) {
    const long total = rows * (K / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / (K / 4);
        const int c = (int)(i - r * (K / 4)) * 4;
        const float4 f = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float v[4] = {f.x, f.y, f.z, f.w};
        u32x2 ph, pm, pl;
        split3(v, ph, pm, pl);
        bf16_t* dst = planes + r * NPL * K + c;
        *reinterpret_cast<u32x2*>(dst) = ph;
        *reinterpret_cast<u32x2*>(dst + K) = pm;
        if (NPL == 3) *reinterpret_cast<u32x2*>(dst + 2 * K) = pl;
    }
}

struct AttnF32Args {
    const float* q; const float* k; const float* v; float* out;
    int B, T, H, ld, ldo;
    float sc;                                                 // scale * log2(e)
    bf16_t* planes; int ldp, pd, npl;                         // split-bf16 route: the context's npl (2 | 3) bf16 planes [M, npl pd] instead of `out`
};
constexpr int AKT = 64, ALD = 65;

__global__ __launch_bounds__(256, 2) void attn_f32_kernel(const AttnF32Args p) {
    __shared__ float Ks[2][AKT][ALD];
    __shared__ float Vs[2][AKT][ALD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hi = lane >> 5;
    const int nqt = (p.T + 127) >> 7;
    int id = blockIdx.x;
    const int qt = id % nqt; id /= nqt;
    const int h = id % p.H;
    const int b = id / p.H;
    const size_t tok0 = (size_t)b * p.T;
    const int qloc = qt * 128 + wave * 32 + lq;
    const float* qrow = p.q + (tok0 + (qloc < p.T ? qloc : p.T - 1)) * p.ld + h * 64;
    float qf[32];                                             // B operand of k-step j: Q[q][2 j + hi]
#pragma unroll
    for (int j = 0; j < 32; ++j) qf[j] = qrow[2 * j + hi];
    f32x16 o[2];
    o[0] = f32x16{}; o[1] = f32x16{};
    float m_run = -INFINITY, l_run = 0.f;
    const int ntile = (p.T + AKT - 1) / AKT;
    // staging: 64 rows x 16 float4 per tile and operand = 4 float4 per thread
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;
    float4 rk[4], rv[4];
    auto fetch = [&](int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = t * AKT + srow + 16 * j;
            const size_t row = tok0 + (key < p.T ? key : p.T - 1);       // keys past the image are masked below: re-read the last row
            rk[j] = *reinterpret_cast<const float4*>(p.k + row * p.ld + h * 64 + sc4);
            rv[j] = *reinterpret_cast<const float4*>(p.v + row * p.ld + h * 64 + sc4);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* kd = &Ks[buf][srow + 16 * j][sc4];
            float* vd = &Vs[buf][srow + 16 * j][sc4];
            kd[0] = rk[j].x; kd[1] = rk[j].y; kd[2] = rk[j].z; kd[3] = rk[j].w;
            vd[0] = rv[j].x; vd[1] = rv[j].y; vd[2] = rv[j].z; vd[3] = rv[j].w;
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    const bool idle = qt * 128 + wave * 32 >= p.T;            // a wave whose 32 rows are all past the sequence only stages
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntile) fetch(t + 1);
        if (!idle) {
            f32x16 s[2];
            s[0] = f32x16{}; s[1] = f32x16{};
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][lq][2 * j + hi], qf[j], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][32 + lq][2 * j + hi], qf[j], s[1], 0, 0, 0);
            }
            if ((t + 1) * AKT > p.T) {                            // the image's last tile (uniform)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * AKT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.T) s[kb][r] = -INFINITY;
            }
            float mloc = fmaxf(s[0][0], s[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(s[0][r], s[1][r]));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m_run, mloc);               // finite: every tile holds at least one valid key
            const float msc = m_new * p.sc;
            const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, p.sc, -msc));
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], p.sc, -msc));
                    s[kb][r] = pv;
                    psum += pv;
                }
            l_run = __builtin_fmaf(l_run, alpha, psum);
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            // O^T[d][q] += sum over keys V[key][d] P[q][key]: k-step = (key block kb, register r), lane half hi supplies key(r, hi)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[buf][key][lq], s[kb][r], o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[buf][key][32 + lq], s[kb][r], o[1], 0, 0, 0);
                }
        }
        if (t + 1 < ntile) stash(buf ^ 1);                      // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    if (idle || qloc >= p.T) return;
    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_run;
    // lane holds O[q][d = dt * 32 + (r & 3) + 8 (r >> 2) + 4 hi]
    if (p.planes) {
        bf16_t* prow = p.planes + (tok0 + qloc) * p.ldp + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float v[4] = {o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv, o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv};
                u32x2 ph, pm, pl;
                split3(v, ph, pm, pl);
                bf16_t* dst = prow + dt * 32 + rg * 8 + hi * 4;
                *reinterpret_cast<u32x2*>(dst) = ph;
                *reinterpret_cast<u32x2*>(dst + p.pd) = pm;
                if (p.npl == 3) *reinterpret_cast<u32x2*>(dst + 2 * p.pd) = pl;
            }
        return;
    }
    float* orow = p.out + (tok0 + qloc) * p.ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 v4 = {o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv, o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv};
            *reinterpret_cast<float4*>(orow + dt * 32 + rg * 8 + hi * 4) = v4;
        }
}

// The same attention with both contractions on the bf16 matrix pipe at fp32 accuracy (split-bf16, see gemm_bf16_v5.hip EPI_F32X): fp32
// Q | K | V in, fp32 context (or its three bf16 planes) out, fp32 softmax.
//   * staging: K / V tiles of 64 keys are fetched as fp32 into registers one tile ahead (under the current tile's matrix work), split into
//     hi | mid | lo bf16 planes and written to LDS after the tile's last read: K planes [64 keys][64 d], V planes TRANSPOSED [64 d][64 keys]
//     (a thread fetches 16 consecutive keys of ONE d, lanes along d: coalesced rows, and writes 2 x 16 bytes per plane; the keys of a
//     16-block are stored as 0-3, 8-11 | 4-7, 12-15 so that the 8 keys a lane needs for one MFMA k-slice are 16 contiguous bytes - the
//     perm16 order of attention.hip); both are 128-byte rows with the 16-byte slot XOR-swizzled by (row >> 1) & 7, conflict-free for
//     ds_read_b128;
//   * S^T = sum over the NP significant plane pairs of K_i Q_j^T (v_mfma_f32_32x32x16_bf16, fp32 accumulate, smallest terms first), Q planes
//     in registers for the whole kernel; O^T += sum over the same pairs of V_i^T P_j^T with P split in registers after the fp32 softmax.
//   NP = 6 (three planes, fp32-equivalent): 96 MFMAs of 32 cycles per 64-key tile instead of 128 exact-fp32 MFMAs of 64 cycles;
//   NP = 3 / 4 (two planes = 16 significand bits per operand): 48 / 64 MFMAs, two thirds of the split work and of the LDS.
constexpr int XPL = 64 * 64 * 2;                               // one plane of one tile: 8 KB
template <int NP> struct SplitPairs;                          // (plane of the LDS operand, plane of the register operand), smallest terms first
template <> struct SplitPairs<6> { static constexpr int A[6] = {2, 0, 1, 1, 0, 0}, B[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct SplitPairs<4> { static constexpr int A[4] = {1, 1, 0, 0}, B[4] = {1, 0, 1, 0}; };
template <> struct SplitPairs<3> { static constexpr int A[3] = {1, 0, 0}, B[3] = {0, 1, 0}; };

template <int NP>
__global__ __launch_bounds__(256, 2) void attn_f32_split_kernel(const AttnF32Args p) {
    constexpr int NPL = NP == 6 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) char Kp[NPL * XPL];
    __shared__ __attribute__((aligned(16))) char Vp[NPL * XPL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hi = lane >> 5;
    const int nqt = (p.T + 127) >> 7;
    int id = blockIdx.x;
    const int qt = id % nqt; id /= nqt;
    const int h = id % p.H;
    const int b = id / p.H;
    const size_t tok0 = (size_t)b * p.T;
    const int qloc = qt * 128 + wave * 32 + lq;
    const bool idle = qt * 128 + wave * 32 >= p.T;             // wave-uniform: all 32 query rows past the sequence

    // ---- Q planes (B operand of k-slice kk: q = lane & 31, d = 16 kk + 8 hi .. + 8), kept in registers
    bf16x8 qf[NPL][4];
    {
        const float* qrow = p.q + (tok0 + (qloc < p.T ? qloc : p.T - 1)) * p.ld + h * 64 + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + kk * 16), c = *reinterpret_cast<const float4*>(qrow + kk * 16 + 4);
            const float v0[4] = {a.x, a.y, a.z, a.w}, v1[4] = {c.x, c.y, c.z, c.w};
            u32x2 h0, m0, l0, h1, m1, l1;
            split3(v0, h0, m0, l0);
            split3(v1, h1, m1, l1);
            const u32x4 wh = {h0[0], h0[1], h1[0], h1[1]}, wm = {m0[0], m0[1], m1[0], m1[1]}, wl = {l0[0], l0[1], l1[0], l1[1]};
            qf[0][kk] = __builtin_bit_cast(bf16x8, wh); qf[1][kk] = __builtin_bit_cast(bf16x8, wm);
            if (NPL == 3) qf[NPL - 1][kk] = __builtin_bit_cast(bf16x8, wl);
        }
    }
    // ---- staging maps.  K: thread -> (key = tid >> 2, d = 16 (tid & 3) .. + 16): four float4.  V: thread -> (d = tid & 63, keys 16 (tid >> 6) .. + 16).
    const int kkey = tid >> 2, kd = (tid & 3) * 16;
    const int vd = tid & 63, vg = tid >> 6;
    float4 rk[4];
    float rv[16];
    auto fetch = [&](int t) {
        {
            const int key = t * 64 + kkey;
            const float* src = p.k + (tok0 + (key < p.T ? key : p.T - 1)) * p.ld + h * 64 + kd;     // keys past the image are masked below
#pragma unroll
            for (int j = 0; j < 4; ++j) rk[j] = *reinterpret_cast<const float4*>(src + 4 * j);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = t * 64 + vg * 16 + j;
            rv[j] = p.v[(tok0 + (key < p.T ? key : p.T - 1)) * p.ld + h * 64 + vd];
        }
    };
    auto stash = [&]() {
        {   // K: 16 consecutive d of one key = slots 2 (tid & 3), + 1 of row `kkey`
            u32x2 ph[4], pm[4], pl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float v[4] = {rk[j].x, rk[j].y, rk[j].z, rk[j].w}; split3(v, ph[j], pm[j], pl[j]); }
            const int sw = (kkey >> 1) & 7;
            char* r0 = Kp + kkey * 128 + (((2 * (tid & 3)) ^ sw) << 4);
            char* r1 = Kp + kkey * 128 + (((2 * (tid & 3) + 1) ^ sw) << 4);
            *reinterpret_cast<u32x4*>(r0) = u32x4{ph[0][0], ph[0][1], ph[1][0], ph[1][1]};
            *reinterpret_cast<u32x4*>(r1) = u32x4{ph[2][0], ph[2][1], ph[3][0], ph[3][1]};
            *reinterpret_cast<u32x4*>(r0 + XPL) = u32x4{pm[0][0], pm[0][1], pm[1][0], pm[1][1]};
            *reinterpret_cast<u32x4*>(r1 + XPL) = u32x4{pm[2][0], pm[2][1], pm[3][0], pm[3][1]};
            if (NPL == 3) {
                *reinterpret_cast<u32x4*>(r0 + 2 * XPL) = u32x4{pl[0][0], pl[0][1], pl[1][0], pl[1][1]};
                *reinterpret_cast<u32x4*>(r1 + 2 * XPL) = u32x4{pl[2][0], pl[2][1], pl[3][0], pl[3][1]};
            }
        }
        {   // V^T: keys 16 vg .. + 16 of row d = vd -> slots 2 vg (keys 0-3, 8-11) and 2 vg + 1 (keys 4-7, 12-15)
            u32x2 ph[4], pm[4], pl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float v[4] = {rv[4 * j], rv[4 * j + 1], rv[4 * j + 2], rv[4 * j + 3]}; split3(v, ph[j], pm[j], pl[j]); }
            const int sw = (vd >> 1) & 7;
            char* r0 = Vp + vd * 128 + (((2 * vg) ^ sw) << 4);
            char* r1 = Vp + vd * 128 + (((2 * vg + 1) ^ sw) << 4);
            *reinterpret_cast<u32x4*>(r0) = u32x4{ph[0][0], ph[0][1], ph[2][0], ph[2][1]};
            *reinterpret_cast<u32x4*>(r1) = u32x4{ph[1][0], ph[1][1], ph[3][0], ph[3][1]};
            *reinterpret_cast<u32x4*>(r0 + XPL) = u32x4{pm[0][0], pm[0][1], pm[2][0], pm[2][1]};
            *reinterpret_cast<u32x4*>(r1 + XPL) = u32x4{pm[1][0], pm[1][1], pm[3][0], pm[3][1]};
            if (NPL == 3) {
                *reinterpret_cast<u32x4*>(r0 + 2 * XPL) = u32x4{pl[0][0], pl[0][1], pl[2][0], pl[2][1]};
                *reinterpret_cast<u32x4*>(r1 + 2 * XPL) = u32x4{pl[1][0], pl[1][1], pl[3][0], pl[3][1]};
            }
        }
    };
    // plane pairs (operand from LDS, operand from registers), smallest product terms first
    using PR = SplitPairs<NP>;
    const int rsw = (lq >> 1) & 7, rbase = lq * 128;

    f32x16 o[2];
    o[0] = f32x16{}; o[1] = f32x16{};
    float m_run = -INFINITY, l_run = 0.f;
    const int ntile = (p.T + 63) / 64;
    fetch(0);
    stash();
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile) fetch(t + 1);                       // next tile's fp32 rows in flight under this tile's matrix work
        if (!idle) {
            f32x16 s[2];
            s[0] = f32x16{}; s[1] = f32x16{};
#pragma unroll
            for (int pr = 0; pr < NP; ++pr)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int kt2 = 0; kt2 < 2; ++kt2) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kp + PR::A[pr] * XPL + kt2 * 4096 + rbase + (((2 * kk + hi) ^ rsw) << 4));
                        s[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[PR::B[pr]][kk], s[kt2], 0, 0, 0);
                    }
            if ((t + 1) * 64 > p.T) {                          // last tile: mask the keys past the image
#pragma unroll
                for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * 64 + kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.T) s[kt2][r] = -INFINITY;
            }
            float mloc = fmaxf(s[0][0], s[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(s[0][r], s[1][r]));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m_run, mloc);               // finite: every tile holds at least one valid key
            const float msc = m_new * p.sc;
            const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, p.sc, -msc));
            float psum = 0.f;
            uint32_t pb[NPL][2][8];
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt2][r + e], p.sc, -msc)); psum += v[e]; }
                    u32x2 ph, pm, pl;
                    split3(v, ph, pm, pl);
                    pb[0][kt2][r >> 1] = ph[0]; pb[0][kt2][(r >> 1) + 1] = ph[1];
                    pb[1][kt2][r >> 1] = pm[0]; pb[1][kt2][(r >> 1) + 1] = pm[1];
                    if (NPL == 3) { pb[NPL - 1][kt2][r >> 1] = pl[0]; pb[NPL - 1][kt2][(r >> 1) + 1] = pl[1]; }
                }
            l_run = __builtin_fmaf(l_run, alpha, psum);
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            // O^T += V^T P^T: chunk c = 16 keys; the P operand of plane j = 4 packed words of pb[j][c >> 1], words 4 (c & 1) .. + 4
#pragma unroll
            for (int pr = 0; pr < NP; ++pr)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32x4 w = {pb[PR::B[pr]][c >> 1][4 * (c & 1) + 0], pb[PR::B[pr]][c >> 1][4 * (c & 1) + 1], pb[PR::B[pr]][c >> 1][4 * (c & 1) + 2],
                                     pb[PR::B[pr]][c >> 1][4 * (c & 1) + 3]};
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, w);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vp + PR::A[pr] * XPL + dt * 4096 + rbase + (((2 * c + hi) ^ rsw) << 4));
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                    }
                }
        }
        __syncthreads();                                        // every wave is done reading this tile's planes
        if (t + 1 < ntile) stash();
        __syncthreads();
    }
    if (idle || qloc >= p.T) return;
    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_run;
    // lane holds O[q][d = dt * 32 + (r & 3) + 8 (r >> 2) + 4 hi]
    if (p.planes) {
        bf16_t* prow = p.planes + (tok0 + qloc) * p.ldp + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float v[4] = {o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv, o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv};
                u32x2 ph, pm, pl;
                split3(v, ph, pm, pl);
                bf16_t* dst = prow + dt * 32 + rg * 8 + hi * 4;
                *reinterpret_cast<u32x2*>(dst) = ph;
                *reinterpret_cast<u32x2*>(dst + p.pd) = pm;
                if (NPL == 3) *reinterpret_cast<u32x2*>(dst + 2 * p.pd) = pl;
            }
        return;
    }
    float* orow = p.out + (tok0 + qloc) * p.ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 v4 = {o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv, o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv};
            *reinterpret_cast<float4*>(orow + dt * 32 + rg * 8 + hi * 4) = v4;
        }
}

// LayerNorm over the last dimension, fp32 throughout, two-pass variance (mean first, then the centred squares) - one wave per row
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ y, int ldy, int rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = xr[c] - mean; q += t * t; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    float* yr = y + (size_t)row * ldy;
    for (int c = lane; c < d; c += 64) yr[c] = (xr[c] - mean) * rstd * g[c] + b[c];
}

// the same LayerNorm writing the NPL bf16 planes of its output [rows, NPL d] (the split-bf16 GEMM's operand) instead of fp32; d % 4 == 0
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_f32_split_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                                  const float* __restrict__ b, bf16_t* __restrict__ planes, int rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = xr[c] - mean; q += t * t; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    bf16_t* pr = planes + (size_t)row * NPL * d;
    for (int c = lane * 4; c < d; c += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + c), gv = *reinterpret_cast<const float4*>(g + c), bv = *reinterpret_cast<const float4*>(b + c);
        const float v[4] = {(xv.x - mean) * rstd * gv.x + bv.x, (xv.y - mean) * rstd * gv.y + bv.y, (xv.z - mean) * rstd * gv.z + bv.z,
                            (xv.w - mean) * rstd * gv.w + bv.w};
        u32x2 ph, pm, pl;
        split3(v, ph, pm, pl);
        *reinterpret_cast<u32x2*>(pr + c) = ph;
        *reinterpret_cast<u32x2*>(pr + d + c) = pm;
        if (NPL == 3) *reinterpret_cast<u32x2*>(pr + 2 * d + c) = pl;
    }
}

// in-place softmax over `cols` of every row (max-subtracted, expf, one division per element) - one wave per row
__global__ __launch_bounds__(256) void softmax_f32_kernel(float* __restrict__ x, int ld, long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + row * ld;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { const float e = expf(xr[c] - m); xr[c] = e; s += e; }
    s = wave_sum(s);
    for (int c = lane; c < cols; c += 64) xr[c] = xr[c] / s;
}

// patch gather for the patch-embedding GEMM: cols[b * P + p][c * ps * ps + ky * ps + kx] = px[b][c][gy * ps + ky][gx * ps + kx], zero pad to kpad
__global__ void im2col_f32_kernel(const float* __restrict__ px, float* __restrict__ cols, int B, int img, int ps, int kpad) {
    const int grid = img / ps, P = grid * grid, kk = 3 * ps * ps;
    const long total = (long)B * P * kpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % kpad);
        const long bp = i / kpad;
        float v = 0.f;
        if (k < kk) {
            const int b = (int)(bp / P), pi = (int)(bp % P), gy = pi / grid, gx = pi % grid;
            const int c = k / (ps * ps), rem = k % (ps * ps), ky = rem / ps, kx = rem % ps;
            v = px[(((size_t)b * 3 + c) * img + gy * ps + ky) * img + gx * ps + kx];
        }
        cols[i] = v;
    }
}

// token rows of the embedding: x[b, cls_off + p] = patches[b * P + p] + pos[cls_off + p];  x[b, 0] = cls + pos[0] (CLS towers)
__global__ void embed_finish_f32_kernel(const float* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
                                        float* __restrict__ x, int B, int T, int P, int d) {
    const int cls_off = T - P;
    const long total = (long)B * T * d;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d);
        const long bt = i / d;
        const int b = (int)(bt / T), t = (int)(bt % T);
        const float pv = pos[(size_t)t * d + c];
        x[i] = (t < cls_off) ? cls[c] + pv : patches[((size_t)b * P + (t - cls_off)) * d + c] + pv;
    }
}

// 3x3 patch gather on channels-last fp32 tokens (pad 1, stride 1): cols[(b, y, x)][(ky * 3 + kx) * C + c] = x[b, y + ky - 1, x + kx - 1, c] or 0
__global__ void im2col3x3_f32_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H, int W, int C) {
    const long total = (long)B * H * W * 9 * (C / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (C / 4));
        long r = i / (C / 4);
        const int tap = (int)(r % 9); r /= 9;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + sy) * W + sx) * C + 4 * c4);
        *reinterpret_cast<float4*>(cols + ((((size_t)b * H + yy) * W + xx) * 9 + tap) * C + 4 * c4) = v;
    }
}

// nn.GroupNorm on channels-last fp32 tokens [B, HW, C] fused with what follows it in a detectron2-style bottleneck block:
//   y = alpha * act( GN(x) * gamma + beta + resid ) (+ y when accumulate)          one workgroup per (image, group), two-pass variance
__global__ __launch_bounds__(256) void groupnorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ resid, float* __restrict__ y, int HW, int C, int groups, float eps,
                                                            int relu, float alpha, int accumulate) {
    __shared__ float red[8];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups, cpg = C / groups;
    const float* xb = x + (size_t)b * HW * C + g * cpg;
    const int n = HW * cpg;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += xb[(size_t)(i / cpg) * C + i % cpg];
    const float mean = block_sum(s) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float t = xb[(size_t)(i / cpg) * C + i % cpg] - mean; q += t * t; }
    const float rstd = 1.0f / sqrtf(block_sum(q) / (float)n + eps);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = g * cpg + i % cpg;
        const size_t off = ((size_t)b * HW + i / cpg) * C + c;
        float v = (x[off] - mean) * rstd * gamma[c] + beta[c];
        if (resid) v += resid[off];
        if (relu) v = fmaxf(v, 0.f);
        v *= alpha;
        y[off] = accumulate ? y[off] + v : v;
    }
}

int launch_gemm_f32(const GemmF32Args& a, int nb, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || nb <= 0) return 0;
    if ((a.lda & 3) || (a.ldw & 3) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) || (a.sA1 & 3) || (a.sA2 & 3) || (a.sW1 & 3) || (a.sW2 & 3))
        return visrep_set_error(VISREP_ERR_SHAPE, "gemm_f32: operand rows must be 16-byte aligned (pointers, leading dimensions and batch strides % 4 floats)");
    dim3 grid((a.N + FBN - 1) / FBN, (a.M + FBM - 1) / FBM, nb);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "gemm_f32: launch failed");
}

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
struct WsF32 { size_t h, qkv, sc, mlp, total; int lds; };
WsF32 layout_f32(const visrep_vit_config* c, int B) {
    WsF32 w;
    const size_t M = (size_t)B * c->tokens;
    w.lds = (c->tokens + 3) / 4 * 4;
    size_t off = 0;
    w.h = off;   off += up256(M * c->d * 4);
    w.qkv = off; off += up256(M * 3 * c->d * 4);
    w.sc = off;  off += up256((size_t)B * c->heads * c->tokens * w.lds * 4);
    const size_t mlp_b = M * c->mlp * 4;
    const size_t P = (size_t)c->tokens - c->has_cls;
    const size_t emb_b = up256((size_t)B * P * c->kpad * 4) + (size_t)B * P * c->d * 4;    // im2col columns + patch rows alias the MLP buffer
    w.mlp = off; off += up256(mlp_b > emb_b ? mlp_b : emb_b);
    w.total = off;
    return w;
}

// split-bf16 route: planes of the LayerNorm / attention output [M, 3 d] bf16, fp32 Q | K | V [M, 3 d], planes of the MLP hidden [M, 3 mlp] bf16
// (the embedding's im2col columns and patch rows alias it)
struct WsF32X { size_t hp, qkv, mlpp, total; };
WsF32X layout_f32x(const visrep_vit_config* c, int B) {
    WsF32X w;
    const size_t M = (size_t)B * c->tokens;
    size_t off = 0;
    w.hp = off;  off += up256(M * 3 * c->d * 2);
    w.qkv = off; off += up256(M * 3 * c->d * 4);
    const size_t mlp_b = M * 3 * c->mlp * 2;
    const size_t P = (size_t)c->tokens - c->has_cls;
    const size_t emb_b = up256((size_t)B * P * c->kpad * 4) + (size_t)B * P * c->d * 4;
    w.mlpp = off; off += up256(mlp_b > emb_b ? mlp_b : emb_b);
    w.total = off;
    return w;
}

// fp32 GEMM on the bf16 matrix pipe: C = epilogue(A W^T) with A, W given as bf16 planes [rows, npl K] (see GemmArgs::ksplit); products in {3, 4, 6}
int launch_gemm_split(const bf16_t* Ap, const bf16_t* Wp, int M, int N, int K, int products, const float* bias, int act, const float* resid, const float* ls,
                      float* C, int ldc, bf16_t* planes, hipStream_t s) {
    GemmArgs a{};
    if (!visrep_split_tables(products, a.tab_a, a.tab_w)) return visrep_set_error(VISREP_ERR_ARG, "gemm_f32_split: products must be 3, 4 or 6");
    const int npl = visrep_split_planes(products);
    a.A = Ap; a.W = Wp; a.C = reinterpret_cast<bf16_t*>(C); a.bias = bias; a.ls = ls; a.resid32 = resid; a.planes = planes; a.ldp = npl * N; a.out_planes = npl;
    a.M = M; a.N = N; a.K = products * K; a.ksplit = K; a.lda = npl * K; a.ldw = npl * K; a.ldc = ldc; a.epi = EPI_F32X; a.act = act;
    if (!visrep_gemm_v5_supports(a)) return visrep_set_error(VISREP_ERR_SHAPE, "gemm_f32_split: N % 256 == 0 and K % 64 == 0");
    return visrep_gemm_v5_dispatch(a, s);
}

bool split_route_supported(const visrep_vit_config* c) {
    return c->d % 256 == 0 && c->mlp % 256 == 0 && c->d % 64 == 0 && c->mlp % 64 == 0 && (c->d / c->heads) == 64;
}

}  // namespace

#define VR_TRY(x) do { const int rc_ = (x); if (rc_) return rc_; } while (0)

extern "C" int visrep_split_bf16_planes(const float* x, int ldx, long rows, int K, int nplanes, void* planes, void* stream) {
    if (!x || !planes) return visrep_set_error(VISREP_ERR_ARG, "split_bf16_planes: null pointer");
    if (nplanes != 2 && nplanes != 3) return visrep_set_error(VISREP_ERR_ARG, "split_bf16_planes: nplanes must be 2 or 3");
    if (rows <= 0) return 0;
    if (K <= 0 || (K & 3) || (ldx & 3) || ((uintptr_t)x & 15) || ((uintptr_t)planes & 7))
        return visrep_set_error(VISREP_ERR_SHAPE, "split_bf16_planes: K and ldx must be multiples of 4, x 16-byte aligned");
    const long total = rows * (K / 4);
    const dim3 grid((unsigned)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16));
    if (nplanes == 3) hipLaunchKernelGGL(split_bf16_planes_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, rows, K, (bf16_t*)planes);
    else hipLaunchKernelGGL(split_bf16_planes_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, rows, K, (bf16_t*)planes);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "split_bf16_planes: launch failed");
}

extern "C" int visrep_gemm_f32_split(const void* a_planes, const void* w_planes, int M, int N, int K, int products, const float* bias, int act,
                                     const float* resid, const float* ls, float* C, int ldc, void* out_planes, void* stream) {
    if (!a_planes || !w_planes || (!C && !out_planes)) return visrep_set_error(VISREP_ERR_ARG, "gemm_f32_split: null pointer");
    if (M <= 0) return 0;
    if (C && (ldc < N || (ldc & 3))) return visrep_set_error(VISREP_ERR_SHAPE, "gemm_f32_split: ldc >= N, % 4 == 0");
    return launch_gemm_split((const bf16_t*)a_planes, (const bf16_t*)w_planes, M, N, K, products, bias, act, resid, ls, C, C ? ldc : N,
                             (bf16_t*)out_planes, (hipStream_t)stream);
}

extern "C" int visrep_debug_f32_attention(int mask) {          // per-thread diagnostic, see t_visrep_f32_attention_dbg; returns the previous mask
    const int old = t_visrep_f32_attention_dbg;
    t_visrep_f32_attention_dbg = mask;
    return old;
}

extern "C" int visrep_vit_f32_split_supported(const visrep_vit_config* cfg) { return cfg && split_route_supported(cfg) ? 1 : 0; }

extern "C" int visrep_gemm_f32(const float* A, int lda, const float* W, int ldw, int w_kn, const float* bias, float* C, int ldc, int M, int N, int K,
                               int epilogue, int act, const float* resid, const float* ls, float alpha, int nb1, int nb2, const long* strides6,
                               void* stream) {
    if (!A || !W || !C) return visrep_set_error(VISREP_ERR_ARG, "gemm_f32: null pointer");
    if (epilogue != VISREP_EPI_BIAS && epilogue != VISREP_EPI_ACT && epilogue != VISREP_EPI_RESID)
        return visrep_set_error(VISREP_ERR_ARG, "gemm_f32: epilogue must be BIAS, ACT or RESID");
    if (epilogue == VISREP_EPI_RESID && !resid) return visrep_set_error(VISREP_ERR_ARG, "gemm_f32: EPI_RESID needs resid");
    if (nb1 < 1 || nb2 < 1 || (nb1 * nb2 > 1 && !strides6)) return visrep_set_error(VISREP_ERR_ARG, "gemm_f32: batch counts >= 1; strides needed when batched");
    GemmF32Args a{};
    a.A = A; a.W = W; a.C = C; a.bias = bias; a.resid = resid; a.ls = ls;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.w_kn = w_kn; a.epi = epilogue; a.act = act; a.alpha = alpha; a.nb2 = nb2;
    if (strides6) { a.sA1 = strides6[0]; a.sA2 = strides6[1]; a.sW1 = strides6[2]; a.sW2 = strides6[3]; a.sC1 = strides6[4]; a.sC2 = strides6[5]; }
    return launch_gemm_f32(a, nb1 * nb2, (hipStream_t)stream);
}

// Gram matrices of image pairs from a bank of position-major maps: G[z] = bank[idx1[z]] bank[idx2[z]]^T  ([PP, C] x [PP, C]^T -> [PP, PP])
extern "C" int visrep_gram_pairs_f32(const float* bank, const int* idx1, const int* idx2, int n_pairs, int PP, int C, float* gram, void* stream) {
    if (!bank || !idx1 || !idx2 || !gram) return visrep_set_error(VISREP_ERR_ARG, "gram_pairs_f32: null pointer");
    if (n_pairs <= 0) return 0;
    if (PP <= 0 || C <= 0 || (C & 3) || (PP & 3)) return visrep_set_error(VISREP_ERR_SHAPE, "gram_pairs_f32: PP and C must be multiples of 4");
    GemmF32Args a{};
    a.A = bank; a.W = bank; a.C = gram; a.M = PP; a.N = PP; a.K = C; a.lda = C; a.ldw = C; a.ldc = PP; a.epi = EPI_BIAS; a.alpha = 1.f; a.nb2 = 1;
    a.sA1 = (long)PP * C; a.sW1 = (long)PP * C; a.sC1 = (long)PP * PP; a.idxA = idx1; a.idxW = idx2;
    return launch_gemm_f32(a, n_pairs, (hipStream_t)stream);
}

extern "C" int visrep_im2col3x3_f32(const float* x, float* cols, int B, int H, int W, int C, void* stream) {
    if (!x || !cols) return visrep_set_error(VISREP_ERR_ARG, "im2col3x3_f32: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return visrep_set_error(VISREP_ERR_SHAPE, "im2col3x3_f32: C must be a positive multiple of 4");
    hipLaunchKernelGGL(im2col3x3_f32_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, x, cols, B, H, W, C);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "im2col3x3_f32: launch failed");
}

extern "C" int visrep_groupnorm_f32(const float* x, const float* gamma, const float* beta, const float* resid, float* y, int B, int HW, int C, int groups,
                                    float eps, int relu, float alpha, int accumulate, void* stream) {
    if (!x || !gamma || !beta || !y) return visrep_set_error(VISREP_ERR_ARG, "groupnorm_f32: null pointer");
    if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm_f32: C must be a multiple of groups");
    hipLaunchKernelGGL(groupnorm_f32_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, resid, y, HW, C, groups, eps, relu, alpha, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "groupnorm_f32: launch failed");
}

extern "C" int visrep_layernorm_f32(const float* x, int ldx, const float* g, const float* b, float* y, int ldy, int rows, int d, float eps, void* stream) {
    if (!x || !g || !b || !y) return visrep_set_error(VISREP_ERR_ARG, "layernorm_f32: null pointer");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, g, b, y, ldy, rows, d, eps);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "layernorm_f32: launch failed");
}

extern "C" int visrep_softmax_rows_f32(float* x, int ld, long rows, int cols, void* stream) {
    if (!x) return visrep_set_error(VISREP_ERR_ARG, "softmax_rows_f32: null pointer");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(softmax_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "softmax_rows_f32: launch failed");
}

extern "C" size_t visrep_vit_f32_workspace_bytes(const visrep_vit_config* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    const size_t a = layout_f32(cfg, B).total, b = layout_f32x(cfg, B).total;     // one workspace serves both routes
    return a > b ? a : b;
}

// The composed fp32 forward: same structure as visrep_vit_forward (HF CLIPVisionTransformer / Dinov2Model / SiglipVisionTransformer
// up to hidden_states[n_layers]), every tensor fp32.  weights: the visrep_vit_weights struct with fp32 MATRICES (patch_w [d, kpad],
// wqkv [3d, d], wo, w1, w2) - `sqkv` / `s1` (the bf16 path's folded-LayerNorm column sums) are ignored.
extern "C" int visrep_vit_forward_f32(const visrep_vit_config* c, const visrep_vit_weights* w, const float* pixels, float* hidden, int B, int n_layers,
                                      void* workspace, void* stream) {
    if (!c || !w || !pixels || !hidden || !workspace) return visrep_set_error(VISREP_ERR_ARG, "vit_forward_f32: null pointer");
    if (B <= 0) return 0;
    if (n_layers < 0 || n_layers > c->layers) return visrep_set_error(VISREP_ERR_ARG, "vit_forward_f32: n_layers out of range");
    if (c->d % c->heads || (c->d / c->heads) % 4 || c->d % 4 || c->mlp % 4 || c->kpad % 4)
        return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward_f32: d, head width, mlp and kpad must be multiples of 4");
    const int grid = c->image_size / c->patch;
    if (grid * grid + (c->has_cls ? 1 : 0) != c->tokens) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward_f32: tokens != grid^2 + cls");
    hipStream_t s = (hipStream_t)stream;
    const WsF32 L = layout_f32(c, B);
    char* base = (char*)workspace;
    float* x = hidden;
    float* h = (float*)(base + L.h);
    float* qkv = (float*)(base + L.qkv);
    float* sc = (float*)(base + L.sc);
    float* mlp = (float*)(base + L.mlp);
    const int d = c->d, T = c->tokens, P = grid * grid, M = B * T, H = c->heads, dh = d / H;

    // ---- embeddings: conv(k = s = patch) = im2col + GEMM (+ bias), then CLS row / position add
    float* cols = mlp;
    float* prow = (float*)((char*)mlp + up256((size_t)B * P * c->kpad * 4));
    hipLaunchKernelGGL(im2col_f32_kernel, dim3(2048), dim3(256), 0, s, pixels, cols, B, c->image_size, c->patch, c->kpad);
    GemmF32Args g{};
    g.A = cols; g.lda = c->kpad; g.W = (const float*)w->patch_w; g.ldw = c->kpad; g.C = prow; g.ldc = d; g.bias = w->patch_b;
    g.M = B * P; g.N = d; g.K = c->kpad; g.epi = EPI_BIAS; g.alpha = 1.f; g.nb2 = 1;
    VR_TRY(launch_gemm_f32(g, 1, s));
    hipLaunchKernelGGL(embed_finish_f32_kernel, dim3(2048), dim3(256), 0, s, prow, w->cls, w->pos, x, B, T, P, d);
    if (hipGetLastError() != hipSuccess) return visrep_set_error(VISREP_ERR_LAUNCH, "vit_forward_f32: embedding launch failed");
    if (c->pre_ln) VR_TRY(visrep_layernorm_f32(x, d, w->pre_ln_g, w->pre_ln_b, x, d, M, d, c->eps, stream));

    const float scale = 1.0f / sqrtf((float)dh);
    for (int l = 0; l < n_layers; ++l) {
        const visrep_vit_layer& W = w->layers[l];
        VR_TRY(visrep_layernorm_f32(x, d, W.ln1_g, W.ln1_b, h, d, M, d, c->eps, stream));
        GemmF32Args a{};
        a.alpha = 1.f; a.nb2 = 1;
        a.A = h; a.lda = d; a.K = d; a.M = M; a.W = (const float*)W.wqkv; a.ldw = d; a.N = 3 * d; a.C = qkv; a.ldc = 3 * d; a.bias = W.bqkv; a.epi = EPI_BIAS;
        VR_TRY(launch_gemm_f32(a, 1, s));
        if (dh == 64 && !(t_visrep_f32_attention_dbg & 1)) {
            // fused flash-style fp32 attention: scores stay in registers (attn_f32_kernel)
            AttnF32Args at{qkv, qkv + d, qkv + 2 * d, h, B, T, H, 3 * d, d, scale * 1.4426950408889634f};
            hipLaunchKernelGGL(attn_f32_kernel, dim3(((T + 127) / 128) * H * B), dim3(256), 0, s, at);
            if (hipGetLastError() != hipSuccess) return visrep_set_error(VISREP_ERR_LAUNCH, "vit_forward_f32: attention launch failed");
        } else {
        // scores[b, head] = scale * Q K^T   (batched over image b1 and head b2)
            GemmF32Args q{};
            q.A = qkv; q.lda = 3 * d; q.W = qkv + d; q.ldw = 3 * d; q.C = sc; q.ldc = L.lds; q.M = T; q.N = T; q.K = dh; q.epi = EPI_BIAS; q.alpha = scale;
            q.nb2 = H; q.sA1 = (long)T * 3 * d; q.sA2 = dh; q.sW1 = (long)T * 3 * d; q.sW2 = dh; q.sC1 = (long)H * T * L.lds; q.sC2 = (long)T * L.lds;
            VR_TRY(launch_gemm_f32(q, B * H, s));
            VR_TRY(visrep_softmax_rows_f32(sc, L.lds, (long)B * H * T, T, stream));
            // context[b, :, head] = P V
            GemmF32Args v{};
            v.A = sc; v.lda = L.lds; v.W = qkv + 2 * d; v.ldw = 3 * d; v.w_kn = 1; v.C = h; v.ldc = d; v.M = T; v.N = dh; v.K = T; v.epi = EPI_BIAS; v.alpha = 1.f;
            v.nb2 = H; v.sA1 = (long)H * T * L.lds; v.sA2 = (long)T * L.lds; v.sW1 = (long)T * 3 * d; v.sW2 = dh; v.sC1 = (long)T * d; v.sC2 = dh;
            VR_TRY(launch_gemm_f32(v, B * H, s));
        }
        // out projection + LayerScale + residual (in place on x)
        a.A = h; a.W = (const float*)W.wo; a.N = d; a.C = x; a.ldc = d; a.bias = W.bo; a.epi = EPI_RESID; a.resid = x; a.ls = W.ls1;
        VR_TRY(launch_gemm_f32(a, 1, s));
        VR_TRY(visrep_layernorm_f32(x, d, W.ln2_g, W.ln2_b, h, d, M, d, c->eps, stream));
        GemmF32Args f{};
        f.alpha = 1.f; f.nb2 = 1;
        f.A = h; f.lda = d; f.K = d; f.M = M; f.W = (const float*)W.w1; f.ldw = d; f.N = c->mlp; f.C = mlp; f.ldc = c->mlp; f.bias = W.b1; f.epi = EPI_ACT; f.act = c->act;
        VR_TRY(launch_gemm_f32(f, 1, s));
        f.A = mlp; f.lda = c->mlp; f.K = c->mlp; f.W = (const float*)W.w2; f.ldw = c->mlp; f.N = d; f.C = x; f.ldc = d; f.bias = W.b2; f.epi = EPI_RESID; f.act = 0;
        f.resid = x; f.ls = W.ls2;
        VR_TRY(launch_gemm_f32(f, 1, s));
    }
    return 0;
}


// The same forward with every projection on the bf16 matrix pipe at fp32 accuracy (16x the exact-fp32 MFMA rate for 6x the products):
// LayerNorm and the fused fp32 attention write the three bf16 planes of their outputs, the projections are EPI_F32X GEMMs over the six
// significant plane pairs (error ~1e-7 relative per product sum, below the fp32 rounding of the native route), bias / activation /
// LayerScale / residual stay fp32 in their epilogues, the MLP hidden only ever exists as planes.  wsplit: a visrep_vit_weights whose
// wqkv / wo / w1 / w2 point to the bf16 plane triples [N, 3 K] of the fp32 matrices (visrep_split_bf16x3); its other fields are ignored.
// Towers whose shapes the 256 x 256 kernel does not take (d or mlp not a multiple of 256, head width != 64:
// visrep_vit_f32_split_supported) must use visrep_vit_forward_f32.
extern "C" int visrep_vit_forward_f32_split(const visrep_vit_config* c, const visrep_vit_weights* w, const visrep_vit_weights* wsplit, int products,
                                            const float* pixels, float* hidden, int B, int n_layers, void* workspace, void* stream) {
    if (!c || !w || !wsplit || !pixels || !hidden || !workspace) return visrep_set_error(VISREP_ERR_ARG, "vit_forward_f32_split: null pointer");
    if (products != 3 && products != 4 && products != 6) return visrep_set_error(VISREP_ERR_ARG, "vit_forward_f32_split: products must be 3, 4 or 6");
    const int npl = visrep_split_planes(products);
    if (B <= 0) return 0;
    if (n_layers < 0 || n_layers > c->layers) return visrep_set_error(VISREP_ERR_ARG, "vit_forward_f32_split: n_layers out of range");
    if (!split_route_supported(c)) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward_f32_split: d and mlp must be multiples of 256, head width 64");
    const int grid = c->image_size / c->patch;
    if (grid * grid + (c->has_cls ? 1 : 0) != c->tokens) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward_f32_split: tokens != grid^2 + cls");
    hipStream_t s = (hipStream_t)stream;
    const WsF32X L = layout_f32x(c, B);
    char* base = (char*)workspace;
    float* x = hidden;
    bf16_t* hp = (bf16_t*)(base + L.hp);
    float* qkv = (float*)(base + L.qkv);
    bf16_t* mlpp = (bf16_t*)(base + L.mlpp);
    const int d = c->d, T = c->tokens, P = grid * grid, M = B * T, H = c->heads;

    // ---- embeddings: exact-fp32 route (0.3 % of the FLOP)
    float* cols = (float*)mlpp;
    float* prow = (float*)((char*)mlpp + up256((size_t)B * P * c->kpad * 4));
    hipLaunchKernelGGL(im2col_f32_kernel, dim3(2048), dim3(256), 0, s, pixels, cols, B, c->image_size, c->patch, c->kpad);
    GemmF32Args g{};
    g.A = cols; g.lda = c->kpad; g.W = (const float*)w->patch_w; g.ldw = c->kpad; g.C = prow; g.ldc = d; g.bias = w->patch_b;
    g.M = B * P; g.N = d; g.K = c->kpad; g.epi = EPI_BIAS; g.alpha = 1.f; g.nb2 = 1;
    VR_TRY(launch_gemm_f32(g, 1, s));
    hipLaunchKernelGGL(embed_finish_f32_kernel, dim3(2048), dim3(256), 0, s, prow, w->cls, w->pos, x, B, T, P, d);
    if (hipGetLastError() != hipSuccess) return visrep_set_error(VISREP_ERR_LAUNCH, "vit_forward_f32_split: embedding launch failed");
    if (c->pre_ln) VR_TRY(visrep_layernorm_f32(x, d, w->pre_ln_g, w->pre_ln_b, x, d, M, d, c->eps, stream));

    const float scale = 1.0f / sqrtf(64.0f);
    auto ln_split = [&](const float* gma, const float* bta) {
        if (npl == 3) hipLaunchKernelGGL(layernorm_f32_split_kernel<3>, dim3((M + 3) / 4), dim3(256), 0, s, x, d, gma, bta, hp, M, d, c->eps);
        else hipLaunchKernelGGL(layernorm_f32_split_kernel<2>, dim3((M + 3) / 4), dim3(256), 0, s, x, d, gma, bta, hp, M, d, c->eps);
        return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "vit_forward_f32_split: layernorm launch failed");
    };
    for (int l = 0; l < n_layers; ++l) {
        const visrep_vit_layer& W = w->layers[l];
        const visrep_vit_layer& S = wsplit->layers[l];
        VR_TRY(ln_split(W.ln1_g, W.ln1_b));
        VR_TRY(launch_gemm_split(hp, (const bf16_t*)S.wqkv, M, 3 * d, d, products, W.bqkv, ACT_NONE, nullptr, nullptr, qkv, 3 * d, nullptr, s));
        AttnF32Args at{qkv, qkv + d, qkv + 2 * d, nullptr, B, T, H, 3 * d, d, scale * 1.4426950408889634f, hp, npl * d, d, npl};
        const dim3 agrid(((T + 127) / 128) * H * B);
        if (t_visrep_f32_attention_dbg & 2) hipLaunchKernelGGL(attn_f32_kernel, agrid, dim3(256), 0, s, at);
        else if (products == 6) hipLaunchKernelGGL(attn_f32_split_kernel<6>, agrid, dim3(256), 0, s, at);
        else if (products == 4) hipLaunchKernelGGL(attn_f32_split_kernel<4>, agrid, dim3(256), 0, s, at);
        else hipLaunchKernelGGL(attn_f32_split_kernel<3>, agrid, dim3(256), 0, s, at);
        if (hipGetLastError() != hipSuccess) return visrep_set_error(VISREP_ERR_LAUNCH, "vit_forward_f32_split: attention launch failed");
        VR_TRY(launch_gemm_split(hp, (const bf16_t*)S.wo, M, d, d, products, W.bo, ACT_NONE, x, W.ls1, x, d, nullptr, s));
        VR_TRY(ln_split(W.ln2_g, W.ln2_b));
        VR_TRY(launch_gemm_split(hp, (const bf16_t*)S.w1, M, c->mlp, d, products, W.b1, c->act, nullptr, nullptr, nullptr, c->mlp, mlpp, s));
        VR_TRY(launch_gemm_split(mlpp, (const bf16_t*)S.w2, M, d, c->mlp, products, W.b2, ACT_NONE, x, W.ls2, x, d, nullptr, s));
    }
    return 0;
}
