// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  HIP only, wave64 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                    // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) short bf16x8;         // one MFMA A/B operand: 8 bf16 = 4 VGPRs
typedef __attribute__((ext_vector_type(4))) float f32x4;          // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;        // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define VR_DEV __device__ __forceinline__

VR_DEV float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
VR_DEV float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
VR_DEV float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// round-to-nearest-even pack; hipcc lowers this to v_cvt_pk_bf16_f32 on gfx950
VR_DEV uint32_t pack_bf16(float lo, float hi) {
    f32x2 v = {lo, hi};
    bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<uint32_t*>(&h);
}
VR_DEV bf16_t f2bf(float f) { return (bf16_t)(pack_bf16(f, 0.f) & 0xffffu); }

// Async global -> LDS copy, 16 B per lane.  The LDS destination is the wave-uniform base `lds_wave_base`
// plus lane*16 (hardware adds the lane offset); the global source address is per lane.
VR_DEV void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS tile format shared by every MFMA kernel here: rows of 64 bf16 (128 B = eight 16-B slots); the slot index is
// XOR-swizzled with (row>>1)&7 so that a ds_read_b128 of {16 or 32 consecutive rows, one logical slot} is
// bank-conflict free (each 16-lane service group of ds_read_b128 then touches 16 distinct 16-B slots of the
// 256-B bank row).  glds16 writes lane-linear, so the swizzle is applied to the per-lane GLOBAL source address
// (lane that lands in physical slot p fetches logical slot p ^ swz) and again on the read side.
VR_DEV int swz_slot(int row, int slot) { return slot ^ ((row >> 1) & 7); }

VR_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
VR_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Bijective XCD-aware block remap: consecutive logical ids land on the same XCD (blocks are dispatched
// round-robin over the 8 XCDs), so neighbouring tiles share that XCD's L2.  Speed only, never correctness.
VR_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, i = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

enum { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU_ERF = 2, ACT_GELU_TANH = 3 };

// Exact GELU x * Phi(x) without libm's erff (a long branchy polynomial: measured 1.9 ms vs 1.26 ms for the fc1 GEMM with
// QuickGELU): Phi from Abramowitz-Stegun 7.1.26, erfc(z) = poly(t) * exp(-z^2), t = 1 / (1 + p z), |error| <= 1.5e-7 - one
// v_rcp, one v_exp and eight FMAs; the lower tail is computed directly (no 1 - erf cancellation).
VR_DEV float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float half_erfc = 0.5f * p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // 0.5 * erfc(|x| / sqrt 2)
    return x * (x < 0.f ? half_erfc : 1.0f - half_erfc);
}

VR_DEV float apply_act(float x, int act) {
    switch (act) {
        case ACT_QUICK_GELU: return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));   // x*sigmoid(1.702x); 1.702*log2(e)
        case ACT_GELU_ERF: return gelu_erf(x);
        case ACT_GELU_TANH: {   // 0.5 x (1 + tanh u) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3); 2 * sqrt(2/pi) * log2(e) = 2.3022082
            const float v = 2.302208198f * __builtin_fmaf(0.044715f * x * x, x, x);
            return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v));
        }
        default: return x;
    }
}
