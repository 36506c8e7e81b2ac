// Internal (non-ABI) declarations shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/visrep.h"

typedef unsigned short bf16_t;

enum { EPI_BIAS = 0, EPI_ACT = 1, EPI_RESID = 2, EPI_VT = 3, EPI_PATCH = 4, EPI_F32 = 5, EPI_F32X = 6 };

struct GemmArgs {
    const bf16_t* A;      // [M, K] row-major, leading dimension lda (elements)
    const bf16_t* W;      // [N, K] row-major (nn.Linear weight), leading dimension ldw
    bf16_t* C;            // output, leading dimension ldc (fp32 when epi == EPI_F32)
    const float* bias;    // [N] or null
    const bf16_t* resid;  // EPI_RESID: residual rows (may alias C)
    const float* ls;      // EPI_RESID: optional LayerScale gamma [N]
    const float* pos;     // EPI_PATCH: position embedding [tokens, N] fp32
    // LayerNorm folded into the GEMM (EPI_BIAS / EPI_ACT / EPI_VT): A is the RAW residual stream, W = gamma o W_orig,
    // C = rstd[m] * (A W^T - mean[m] * s[n]) + bias'[n] with ln_rt[m] = (rstd, -mean * rstd), ln_s[n] = sum_k W[n, k]
    const float2* ln_rt;  // [M] per-row statistics (visrep_layernorm_stats) or null
    const float* ln_s;    // [N]
    // EPI_RESID that also produces the LayerNorm statistics of its OUTPUT rows (the next block's folded LayerNorm):
    // stat_rt[m] = (rstd, -mean * rstd) over the N output columns.  The 256x256 kernel (v2) writes per-wave-tile partial sums to
    // stat_partial[m * stat_slots + slot] from its epilogue and ln_stats_finalize reduces them in slot order; every other route
    // (split-K, 128x128 tail rows, v3) runs the read-only statistics pass on the rows it produced.  Set by visrep_vit_forward.
    float2* stat_rt;      // [M] or null
    float2* stat_partial; // [M, N / 64] scratch (v2 slot width = 64 columns)
    int stat_slots;       // filled in by the dispatcher
    float stat_eps;
    int M, N, K, lda, ldw, ldc;
    int epi, act;
    int patches, tokens, cls_off;   // EPI_PATCH row remap
    int kslice;                     // split-K (v1, EPI_F32 only): blockIdx.y = slice, K elements per slice; 0 = no split
    // implicit 3x3 convolution (v1 kernel, conv = 1): A is the channels-last activation [B, cH, cW, cC] and the A tile of
    // K-tile (tap, c0) is gathered on the fly: row m = (b, oy, ox) reads x[b, (oy*cstride+ky-cpad)>>cup, (ox*cstride+kx-cpad)>>cup, c0..]
    int conv, cH, cW, cC, cHo, cWo, cstride, cpad, cup;
    // EPI_F32X (variant 5 only): an fp32 GEMM on the bf16 matrix pipe.  A and W hold the three bf16 planes (hi | mid | lo, x = hi + mid + lo
    // to 24 bits) of fp32 matrices side by side: A [M, 3 ksplit], W [N, 3 ksplit]; K = 6 ksplit walks the six plane pairs whose product
    // terms are >= 2^-24 of the result: (hi, hi) (hi, mid) (hi, lo) (mid, hi) (mid, mid) (lo, hi).  Epilogue in fp32:
    // v = act(acc + bias); v = resid32 + ls * v (if resid32); C (fp32, may be null) and / or the three planes of v -> planes [M, 3 N].
    int ksplit;
    const float* resid32;
    bf16_t* planes;
    int ldp;
    unsigned long long* dbg_buf;    // timing-only: per-segment cycle sums (VISREP_GEMM_ABLATE builds)
    int dbg;                        // timing-only ablation mask for the v2 kernel (1 = no MFMA, 2 = no LDS-DMA, 4 = no ds_read); 0 in production
};

int visrep_gemm_dispatch(const GemmArgs& a, hipStream_t s);
int visrep_ln_stats_finalize(const float2* partial, int slots, float2* rt, int rows, int d, float eps, hipStream_t s);
bool visrep_gemm_v2_supports(const GemmArgs& a);
int visrep_gemm_v2_dispatch(const GemmArgs& a, hipStream_t s);
bool visrep_gemm_v4_supports(const GemmArgs& a);
int visrep_gemm_v4_dispatch(const GemmArgs& a, hipStream_t s);
bool visrep_gemm_v5_supports(const GemmArgs& a);
int visrep_gemm_v5_dispatch(const GemmArgs& a, hipStream_t s);
bool visrep_gemm_v3_supports(const GemmArgs& a);
int visrep_gemm_v3_dispatch(const GemmArgs& a, hipStream_t s);
extern int g_visrep_gemm_dbg;
extern unsigned long long* g_visrep_gemm_dbg_buf;
extern int g_visrep_gemm_variant;   // 1 = 128x128 kernel, 2 / 3 / 5 = 256x256 persistent ping-pong kernels (when N % 256 == 0), 4 = 4-wave stream; default 5
int visrep_attention_ab_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* out, int ldo,
                               int B, int Tq, int Tk, int H, int kv_shared, int causal, float scale, hipStream_t st);
extern int g_visrep_attn_variant;
int visrep_set_error(int code, const char* msg);
constexpr int VISREP_MAX_DEVICES = 16;
extern void* g_visrep_scratch[VISREP_MAX_DEVICES];       // caller-owned device scratch (visrep_set_scratch), per device: split-K partial sums
extern size_t g_visrep_scratch_bytes[VISREP_MAX_DEVICES];
