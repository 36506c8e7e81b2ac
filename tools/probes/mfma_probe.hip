// Micro-benchmark: matrix-pipe cadence of the block structures used by the GEMM kernels (no memory traffic at all).
//   V0 free-running: 8 waves (2 per SIMD) each issue 16-MFMA bursts back to back, no barriers
//   V1 lockstep:     8 waves, s_barrier after every burst
//   V2 ping-pong:    two groups of 4 waves skewed by one barrier; a wave alternates {burst | barrier | idle | barrier}
//   V3 ping-pong with a ~200-cycle VALU/SALU filler in the idle segment (stands in for the load segment)
// MF = 0: v_mfma_f32_32x32x16_bf16 (16 per burst), MF = 1: v_mfma_f32_16x16x32_bf16 (32 per burst: same FLOPs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void bar() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

template <int V, int MF, int PRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int rnd, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = bf16x8{(short)(lane + i), 1, 2, 3, 4, 5, 6, 7}; b[i] = bf16x8{(short)(lane * 3 + i), 1, 2, 3, 4, 5, 6, 7}; }
    if (rnd) {      // pseudo-random bf16, |x| in [0.25, 4): sign + biased exponent 0x7d..0x80 + random mantissa (until round 5 the exponent field was 0x3c..0x3f = 1e-20-sized values whose products underflow: an optimistic "random" figure)
        unsigned st = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ 0x9e3779b9u;
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 8; ++e) {
                st = st * 1664525u + 1013904223u; unsigned r = st >> 8;
                a[i][e] = (short)(((r & 1) << 15) | ((0x7d + ((r >> 1) & 3)) << 7) | ((r >> 3) & 0x7f));
                st = st * 1664525u + 1013904223u; r = st >> 8;
                b[i][e] = (short)(((r & 1) << 15) | ((0x7d + ((r >> 1) & 3)) << 7) | ((r >> 3) & 0x7f));
            }
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    f32x16 acc[8]; f32x4 acc4[32];
    for (int i = 0; i < 8; ++i) acc[i] = f32x16{};
    for (int i = 0; i < 32; ++i) acc4[i] = f32x4{0, 0, 0, 0};
    float filler = (float)lane;
    auto burst = [&]() {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        if (MF == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc4[i], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    __shared__ __attribute__((aligned(16))) char lds[65536];
    bf16x8 rd[8];
    auto idle = [&]() {
        if (V >= 4 && V != 7 || V == 7) {      // the GEMM's load segment: NR ds_read_b128 (conflict-free 64-B-row swizzle) + lgkmcnt(0)
            const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * 8192 + (lane & 15) * 64 +
                                  ((((lane >> 4)) ^ ((4 - (((lane & 15) >> 2) & 3)) & 3)) << 4);
            if (V == 4 || V == 6 || V == 7)
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                             "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(rd[0]), "=&v"(rd[1]), "=&v"(rd[2]), "=&v"(rd[3]), "=&v"(rd[4]), "=&v"(rd[5]), "=&v"(rd[6]), "=&v"(rd[7]) : "v"(base));
            else
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(rd[0]), "=&v"(rd[1]), "=&v"(rd[2]), "=&v"(rd[3]) : "v"(base));
            if (V != 6) { a[0] = rd[0]; b[0] = rd[1]; }     // consume (keeps the reads alive); V6: reads only, MFMA operands untouched
            else asm volatile("" ::"v"(rd[0]), "v"(rd[7]));
        }
        if (V == 3) {
#pragma unroll
            for (int i = 0; i < 100; ++i) filler = filler * 1.0001f + 0.5f;
        }
    };
    if (V == 0) {
        for (int it = 0; it < iters; ++it) { burst(); burst(); }
    } else if (V == 1) {
        for (int it = 0; it < iters; ++it) { burst(); bar(); burst(); bar(); }
    } else if (V == 7) {   // load segments only, no MFMA partner: what do 8 reads cost by themselves?
        for (int it = 0; it < iters; ++it) { idle(); bar(); bar(); idle(); bar(); bar(); }
    } else {
        if (grp == 1) bar();
        for (int it = 0; it < iters; ++it) { idle(); bar(); burst(); bar(); idle(); bar(); burst(); bar(); }
        if (grp == 0) bar();
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (clk && threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
    float s = filler;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 32; ++i) s += acc4[i][1];
    if (s == 123.456f) out[0] = s;
}

template <int V, int MF, int PRIO>
void run(const char* name, float* out, int rnd = 0) {
    static unsigned long long* clk = nullptr;
    if (!clk) CHECK(hipMallocManaged(&clk, 16));
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<V, MF, PRIO><<<grid, 512>>>(out, 10, rnd, clk); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<V, MF, PRIO><<<grid, 512>>>(out, iters, rnd, clk);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // FLOPs: per wave per burst 16 * 32*32*16*2 ; bursts per wave: V0,V1: 2*iters (all 8 waves) ; V2,V3: 2*iters (all 8 waves too)
    const double flops = (double)grid * 8 * 2.0 * iters * 16 * 32768.0;
    const double bursts = 2.0 * iters;
    printf("%-44s %s %8.3f ms  %7.1f TFLOP/s   %6.0f ns per burst slot   shader clk %.0f MHz (clock64/wall_clock64 @100MHz)\n", name, rnd ? "random" : "const ", ms,
           flops / ms / 1e9, ms * 1e6 / bursts / (V >= 2 ? 2 : 1), (double)clk[0] / (double)clk[1] * 100.0);
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4));
    run<0, 0, 0>("V0 free-running 32x32x16", out);
    run<0, 1, 0>("V0 free-running 16x16x32", out);
    run<1, 0, 0>("V1 lockstep barrier/burst 32x32x16", out);
    run<1, 1, 0>("V1 lockstep barrier/burst 16x16x32", out);
    run<2, 0, 0>("V2 ping-pong 32x32x16", out);
    run<2, 0, 1>("V2 ping-pong 32x32x16 + setprio", out);
    run<2, 1, 0>("V2 ping-pong 16x16x32 (32/burst)", out);
    run<3, 0, 0>("V3 ping-pong + 200-cyc filler 32x32x16", out);
    run<3, 0, 1>("V3 ping-pong + filler + setprio", out);
    run<0, 0, 0>("V0 free-running 32x32x16", out, 1);
    run<0, 1, 0>("V0 free-running 16x16x32", out, 1);
    run<2, 0, 1>("V2 ping-pong 32x32x16 + setprio", out, 1);
    run<2, 1, 0>("V2 ping-pong 16x16x32 (32/burst)", out, 1);
    run<3, 0, 1>("V3 ping-pong + filler + setprio", out, 1);
    run<4, 1, 1>("V4 ping-pong + 8 ds_read_b128 in L (16x16x32)", out, 1);
    run<5, 1, 1>("V5 ping-pong + 4 ds_read_b128 in L (16x16x32)", out, 1);
    run<4, 0, 1>("V4 ping-pong + 8 ds_read_b128 in L (32x32x16)", out, 1);
    run<7, 1, 0>("V7 8 ds_read_b128 + 2 barriers only (no MFMA)", out, 1);
    return 0;
}
