// Do LDS-DMA writes (global_load_lds) and fragment reads (ds_read_b128) share one LDS pipe?
// 8 waves per block, 1 block per CU.  MODE 0: waves 0-3 stream LDS-DMA (L2-resident 128-B rows), waves 4-7 idle.
// MODE 1: waves 4-7 issue conflict-free ds_read_b128 bursts, waves 0-3 idle.  MODE 2: both at once.
// MODE 3: waves 0-3 stream with global_load -> VGPR -> ds_write_b128 instead (register staging) while 4-7 read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f4;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ base, int iters, unsigned long long* stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave < 4;
    const char* src = base + (size_t)(blockIdx.x & 7) * (4u << 20) + (size_t)(wave * 64 + (lane >> 3)) * 8192 + (lane & 7) * 16;
    const unsigned long long t0 = clock64();
    if (loader && MODE != 1) {
        for (int it = 0; it < iters; ++it) {
            const size_t ko = (size_t)(it & 63) * 128;
#pragma unroll
            for (int j = 0; j < 8; ++j) {                      // 8 KB per wave per iteration (rows j*8.. of its 64-row slab)
                if (MODE == 3) {
                    const f4 v = *reinterpret_cast<const f4*>(src + (size_t)j * 8 * 8192 + ko);
                    *reinterpret_cast<f4*>(smem + ((it & 1) * 4 + wave) * 8192 + j * 1024 + lane * 16) = v;
                } else {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * 8 * 8192 + ko),
                                                     (__attribute__((address_space(3))) void*)(smem + ((it & 1) * 4 + wave) * 8192 + j * 1024), 16, 0, 0);
                }
            }
            if (MODE != 3) asm volatile("s_waitcnt vmcnt(8)");
        }
        asm volatile("s_waitcnt vmcnt(0)");
    }
    if (!loader && MODE != 0) {
        bf16x8 r[8];
        const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 65536 + (wave - 4) * 16384 + (lane & 31) * 128 +
                           (((lane >> 5) ^ (((lane & 31) >> 1) & 7)) << 4);
        for (int it = 0; it < iters; ++it) {
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\tds_read_b128 %3, %8 offset:12288\n\t"
                         "ds_read_b128 %4, %8 offset:32\n\tds_read_b128 %5, %8 offset:4128\n\tds_read_b128 %6, %8 offset:8224\n\tds_read_b128 %7, %8 offset:12320\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]) : "v"(a));
            asm volatile("" ::"v"(r[0]), "v"(r[7]));
        }
    }
    const unsigned long long t1 = clock64();
    if (blockIdx.x == 0 && lane == 0) stats[wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, const char* buf, unsigned long long* st) {
    const int iters = 4000, lds = 131072;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    k<MODE><<<256, 512, lds>>>(buf, 10, st); CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    k<MODE><<<256, 512, lds>>>(buf, iters, st);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double dma_bytes = 256.0 * 4 * iters * 8192.0, rd_bytes = 256.0 * 4 * iters * 8192.0;
    printf("%-52s %7.3f ms | loader wave %8llu cyc (%5.1f cyc / 1-KB load) | reader wave %8llu cyc (%5.1f cyc / ds_read_b128)", name, ms,
           st[0], (double)st[0] / iters / 8, st[4], (double)st[4] / iters / 8);
    if (MODE != 1) printf(" | stream %.1f TB/s", dma_bytes / ms / 1e9);
    if (MODE != 0) printf(" | reads %.1f TB/s", rd_bytes / ms / 1e9);
    printf("\n");
}

int main() {
    char* buf; CHECK(hipMalloc(&buf, 64u << 20)); CHECK(hipMemset(buf, 1, 64u << 20));
    unsigned long long* st; CHECK(hipMallocManaged(&st, 64));
    run<0>("MODE0 LDS-DMA stream only (4 waves)", buf, st);
    run<1>("MODE1 ds_read_b128 only (4 waves)", buf, st);
    run<2>("MODE2 LDS-DMA stream + ds_read_b128", buf, st);
    run<3>("MODE3 global_load+ds_write_b128 stream + ds_read_b128", buf, st);
    return 0;
}
