// Shared fused epilogues of the bf16 GEMM kernels (v1 128x128 and v2 256x256).
//
// Accumulator layout per 16x16 MFMA fragment acc[i][j] (see gemm_bf16.hip):
//   row-major epilogues ("swapped" operands): lane holds C[m = mb + 16 i + fr][n = nb + 16 j + 4 fg + (0..3)]
//   V^T epilogue (plain operand order):       lane holds C[m = mb + 16 i + 4 fg + (0..3)][n = nb + 16 j + fr]
//
// Structure matters more than arithmetic here: every optional vector (bias, LayerScale) is loaded ONCE under a single
// hoisted uniform branch, and residual / position rows are loaded with clamped (always valid) addresses so that no
// load sits behind a per-element branch - hipcc otherwise emits branch + s_waitcnt vmcnt(0) per element, i.e. dozens
// of serialized L2 round trips per tile (and, in v2, a drained LDS-DMA queue).  Only the stores are predicated.
#pragma once
#include "common.h"
#include "visrep_internal.h"

template <int EPI, int NI, int NJ, int ACT, bool EDGE>
VR_DEV void gemm_epilogue_rowmajor_impl(const GemmArgs& p, const f32x4 (&acc)[NI][NJ], int mb, int nb, int fr, int fg) {
    float4 bv[NJ], lv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { bv[j] = float4{0.f, 0.f, 0.f, 0.f}; lv[j] = float4{1.f, 1.f, 1.f, 1.f}; }
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(p.bias + nb + j * 16 + fg * 4);
    }
    if (EPI == EPI_RESID && p.ls) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) lv[j] = *reinterpret_cast<const float4*>(p.ls + nb + j * 16 + fg * 4);
    }
    // rows are processed in groups of RB: all residual / position loads of a group are issued before its first store
    // (in-place residual: loads and stores alias as far as the compiler knows, so source order is the only batching tool)
    constexpr int RB = 4;
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += RB) {
        bool ok[RB];
        size_t orow[RB];
        u32x2 rv[RB][NJ];
        float4 pv[RB][NJ];
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) {
            const int m = mb + (i0 + ii) * 16 + fr;
            ok[ii] = EDGE ? (m < p.M) : true;               // interior tiles: no exec-masked region at all
            const int mc = ok[ii] ? m : p.M - 1;
            orow[ii] = (size_t)mc;
            const float* posrow = nullptr;
            if (EPI == EPI_PATCH) {   // m = b*P + pidx  ->  token row b*T + cls_off + pidx ; add pos[cls_off + pidx]
                const int b = mc / p.patches, pi = mc - b * p.patches;
                orow[ii] = (size_t)b * p.tokens + p.cls_off + pi;
                posrow = p.pos + (size_t)(p.cls_off + pi) * p.N;
            }
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) rv[ii][j] = *reinterpret_cast<const u32x2*>(p.resid + orow[ii] * p.ldc + nb + j * 16 + fg * 4);
            }
            if (EPI == EPI_PATCH) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) pv[ii][j] = *reinterpret_cast<const float4*>(posrow + nb + j * 16 + fg * 4);
            }
        }
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) {
            const int i = i0 + ii;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nb + j * 16 + fg * 4;
                float v0 = acc[i][j][0] + bv[j].x, v1 = acc[i][j][1] + bv[j].y, v2 = acc[i][j][2] + bv[j].z, v3 = acc[i][j][3] + bv[j].w;
                if (EPI == EPI_ACT) {
                    v0 = apply_act(v0, ACT); v1 = apply_act(v1, ACT); v2 = apply_act(v2, ACT); v3 = apply_act(v3, ACT);
                }
                if (EPI == EPI_RESID) {
                    v0 = bf_lo(rv[ii][j][0]) + v0 * lv[j].x; v1 = bf_hi(rv[ii][j][0]) + v1 * lv[j].y;
                    v2 = bf_lo(rv[ii][j][1]) + v2 * lv[j].z; v3 = bf_hi(rv[ii][j][1]) + v3 * lv[j].w;
                }
                if (EPI == EPI_PATCH) { v0 += pv[ii][j].x; v1 += pv[ii][j].y; v2 += pv[ii][j].z; v3 += pv[ii][j].w; }
                if (ok[ii]) {
                    if (EPI == EPI_F32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + orow[ii] * p.ldc + n) = float4{v0, v1, v2, v3};
                    } else {
                        u32x2 o = {pack_bf16(v0, v1), pack_bf16(v2, v3)};
                        *reinterpret_cast<u32x2*>(p.C + orow[ii] * p.ldc + n) = o;
                    }
                }
            }
        }
    }
}

// the activation kind and "does this wave tile touch the M edge" are uniform runtime values: branch ONCE, outside the
// per-element code (an interior tile then has no exec-masked region, so hipcc cannot sink the residual loads into one)
template <int EPI, int NI, int NJ, int ACT>
VR_DEV void gemm_epilogue_rowmajor_act(const GemmArgs& p, const f32x4 (&acc)[NI][NJ], int mb, int nb, int fr, int fg) {
    if (mb + NI * 16 <= p.M) gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, false>(p, acc, mb, nb, fr, fg);
    else gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, true>(p, acc, mb, nb, fr, fg);
}

template <int EPI, int NI, int NJ>
VR_DEV void gemm_epilogue_rowmajor(const GemmArgs& p, const f32x4 (&acc)[NI][NJ], int mb, int nb, int fr, int fg) {
    if (EPI == EPI_ACT) {
        switch (p.act) {
            case ACT_QUICK_GELU: gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_QUICK_GELU>(p, acc, mb, nb, fr, fg); break;
            case ACT_GELU_ERF: gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_GELU_ERF>(p, acc, mb, nb, fr, fg); break;
            case ACT_GELU_TANH: gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_GELU_TANH>(p, acc, mb, nb, fr, fg); break;
            default: gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_NONE>(p, acc, mb, nb, fr, fg); break;
        }
    } else {
        gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_NONE>(p, acc, mb, nb, fr, fg);
    }
}

// V^T scatter: vt[n * ldc + perm16(m)], four consecutive tokens per 8-byte store (layout: see attention.hip)
template <int NI, int NJ>
VR_DEV void gemm_epilogue_vt(const GemmArgs& p, const f32x4 (&acc)[NI][NJ], int mb, int nb, int fr, int fg) {
    float bj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bj[j] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bj[j] = p.bias[nb + j * 16 + fr];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        bf16_t* col = p.C + (size_t)(nb + j * 16 + fr) * p.ldc;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = mb + i * 16 + fg * 4;                                   // multiple of 4
            if (m < p.M) {                                                           // columns >= M are never read unmasked
                const int mp = (m & ~15) | ((((m >> 2) & 1) << 1 | ((m >> 3) & 1)) << 2);   // swap 4-token groups 1 <-> 2
                u32x2 v = {pack_bf16(acc[i][j][0] + bj[j], acc[i][j][1] + bj[j]), pack_bf16(acc[i][j][2] + bj[j], acc[i][j][3] + bj[j])};
                *reinterpret_cast<u32x2*>(col + mp) = v;
            }
        }
    }
}
