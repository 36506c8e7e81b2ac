// Micro-benchmark: what a wave64 VALU instruction costs one SIMD on gfx950, by kind, with 1 / 2 / 4 waves per SIMD, alone and with
// MFMAs in the same wave (1 MFMA + N fillers) or in the partner wave.  Motivates the instruction selection of attention_ab.hip.
//   kinds: 0 v_fma_f32  1 v_pk_fma_f32  2 v_pk_mul_f32  3 v_pk_add_f32  4 v_exp_f32  5 v_cvt_pk_bf16_f32  6 v_max3_f32
//          7 v_exp + v_fma alternating  8 v_mfma_f32_32x32x16_bf16 alone  9 1 MFMA + 8 v_fma  10 1 MFMA + 4 v_fma + 4 v_exp
//          11 1 MFMA + 4 v_pk_fma + 4 v_exp   12 1 MFMA + 8 v_pk_fma
// build: hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip ; run: ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float a[8], b = 1.0001f + lane * 1e-6f, c = 1e-3f;
    f32x2 p[8], pb = {1.0001f, 0.9999f}, pc = {1e-3f, 2e-3f};
    unsigned w[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.5f + 0.01f * i + lane * 1e-3f; p[i] = f32x2{a[i], a[i] * 0.5f}; w[i] = i; }
    f32x16 acc[2] = {f32x16{}, f32x16{}};
    bf16x8 ma = {(short)0x3c00, 1, 2, 3, 4, 5, 6, 7}, mb = {(short)0x3c10, 1, 2, 3, 4, 5, 6, 7};
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(a[i]), "v"(b));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MFMA(j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma, mb, acc[j], 0, 0, 0);
        if (KIND == 0) { R8(FMA) R8(FMA) }
        if (KIND == 1) { R8(PKFMA) R8(PKFMA) }
        if (KIND == 2) { R8(PKMUL) R8(PKMUL) }
        if (KIND == 3) { R8(PKADD) R8(PKADD) }
        if (KIND == 4) { R8(EXP) R8(EXP) }
        if (KIND == 5) { R8(CVT) R8(CVT) }
        if (KIND == 6) { R8(MAX3) R8(MAX3) }
        if (KIND == 7) { EXP(0) FMA(1) EXP(2) FMA(3) EXP(4) FMA(5) EXP(6) FMA(7) EXP(1) FMA(0) EXP(3) FMA(2) EXP(5) FMA(4) EXP(7) FMA(6) }
        if (KIND == 8) { MFMA(0) MFMA(1) }
        if (KIND == 9) { MFMA(0) R8(FMA) MFMA(1) R8(FMA) }
        if (KIND == 10) { MFMA(0) FMA(0) EXP(1) FMA(2) EXP(3) FMA(4) EXP(5) FMA(6) EXP(7) MFMA(1) FMA(1) EXP(0) FMA(3) EXP(2) FMA(5) EXP(4) FMA(7) EXP(6) }
        if (KIND == 11) { MFMA(0) PKFMA(0) EXP(1) PKFMA(2) EXP(3) PKFMA(4) EXP(5) PKFMA(6) EXP(7) MFMA(1) PKFMA(1) EXP(0) PKFMA(3) EXP(2) PKFMA(5) EXP(4) PKFMA(7) EXP(6) }
        if (KIND == 12) { MFMA(0) R8(PKFMA) MFMA(1) R8(PKFMA) }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)w[i];
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char* name, int per_iter_valu, int per_iter_mfma, float* out) {
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {                       // waves per SIMD: blocks of 256 threads (1 wave per SIMD) x wps per CU
        const int grid = 256 * wps;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ns_per_iter_simd = ms * 1e6 / iters / wps;   // time one SIMD spends per loop iteration of ONE wave
        printf("%-28s waves/SIMD %d: %7.2f ns per wave-iteration per SIMD", name, wps, ns_per_iter_simd);
        if (per_iter_valu) printf("  = %5.2f ns per VALU-class op", ns_per_iter_simd / (per_iter_valu + per_iter_mfma));
        printf("\n");
    }
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4));
    run<0>("16 v_fma_f32", 16, 0, out);
    run<1>("16 v_pk_fma_f32", 16, 0, out);
    run<2>("16 v_pk_mul_f32", 16, 0, out);
    run<3>("16 v_pk_add_f32", 16, 0, out);
    run<4>("16 v_exp_f32", 16, 0, out);
    run<5>("16 v_cvt_pk_bf16_f32", 16, 0, out);
    run<6>("16 v_max3_f32", 16, 0, out);
    run<7>("8 v_exp + 8 v_fma", 16, 0, out);
    run<8>("2 mfma 32x32x16", 0, 2, out);
    run<9>("2 x (mfma + 8 v_fma)", 16, 2, out);
    run<10>("2 x (mfma + 4 fma + 4 exp)", 16, 2, out);
    run<11>("2 x (mfma + 4 pk_fma + 4 exp)", 16, 2, out);
    run<12>("2 x (mfma + 8 pk_fma)", 16, 2, out);
    return 0;
}
