// Micro-benchmark: what does the output-store stream of a 256x256 bf16 GEMM tile cost by ITSELF, as a function of the shape of one
// store instruction?  8 waves per CU, wave w owns rows [128 (w>>2), +128) x 128 bytes [(w&3) 128, +128) of the tile, as in gemm_bf16_v2.
//   A: dwordx2, 16 rows x 32 B per instruction (round-1 epilogue)      B: dwordx4, 16 rows x 64 B (round-2 epilogue)
//   C: dwordx4,  8 rows x 128 B per instruction (full 128-B lines)     D: dwordx4, 4 rows x 256 B (wave owns 64 rows x 256 B)
// Build: hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe ; run: ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* __restrict__ out, int ld_bytes, int ntn, int ntiles) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / ntn, tn = t % ntn;
        char* tile = out + (size_t)tm * 256 * ld_bytes + (size_t)tn * 512;
        if (PAT == 0) {          // 16 rows x 32 B, dwordx2: 32 instructions
            char* wb = tile + (size_t)(wave >> 2) * 128 * ld_bytes + (wave & 3) * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    char* p = wb + (size_t)(i * 16 + (lane & 15)) * ld_bytes + c * 32 + (lane >> 4) * 8;
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(f2{v[0], v[1]}) : "memory");
                }
        } else if (PAT == 1) {   // 16 rows x 64 B, dwordx4: 16 instructions
            char* wb = tile + (size_t)(wave >> 2) * 128 * ld_bytes + (wave & 3) * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    char* p = wb + (size_t)(i * 16 + (lane & 15)) * ld_bytes + c * 64 + (lane >> 4) * 16;
                    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
                }
        } else if (PAT == 2) {   // 8 rows x 128 B, dwordx4: 16 instructions
            char* wb = tile + (size_t)(wave >> 2) * 128 * ld_bytes + (wave & 3) * 128;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                char* p = wb + (size_t)(i * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16;
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
            }
        } else if (PAT == 3) {   // 4 rows x 256 B (wave owns 64 rows x 256 B: rows [32 wave.., ) x half the tile width)
            char* wb = tile + (size_t)(wave >> 1) * 64 * ld_bytes + (wave & 1) * 256;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                char* p = wb + (size_t)(i * 4 + (lane >> 4)) * ld_bytes + (lane & 15) * 16;
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
            }
        } else {                 // 2 rows x 512 B (wave owns 32 full tile rows)
            char* wb = tile + (size_t)wave * 32 * ld_bytes;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                char* p = wb + (size_t)(i * 2 + (lane >> 5)) * ld_bytes + (lane & 31) * 16;
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)");
}

template <int PAT>
void run(const char* name, char* buf, int N, int M, int grid) {
    const int ld = N * 2, ntn = N / 256, ntiles = (M / 256) * ntn;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    store_kernel<PAT><<<grid, 512>>>(buf, ld, ntn, ntiles);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) store_kernel<PAT><<<grid, 512>>>(buf, ld, ntn, ntiles);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)ntiles * 131072.0;
    printf("%-28s N %4d grid %3d : %7.3f ms  %6.2f TB/s  %5.2f us/tile/CU  (%.1f B/clk/CU @2.0GHz)\n", name, N, grid, ms, bytes / ms / 1e9,
           ms * 1e3 / (ntiles / (double)grid), bytes / ms / 1e-3 / grid / 2.0e9);
}

int main() {
    const int M = 147456;
    char* buf; CHECK(hipMalloc(&buf, (size_t)M * 4096 * 2));
    for (int N : {4096, 1024}) {
        for (int grid : {256, 64}) {
            run<0>("A dwordx2 16 rows x 32 B", buf, N, M, grid);
            run<1>("B dwordx4 16 rows x 64 B", buf, N, M, grid);
            run<2>("C dwordx4 8 rows x 128 B", buf, N, M, grid);
            run<3>("D dwordx4 4 rows x 256 B", buf, N, M, grid);
            run<4>("E dwordx4 2 rows x 512 B", buf, N, M, grid);
        }
    }
    return 0;
}
