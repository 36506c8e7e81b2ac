// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] * W[N,K]^T )   (nn.Linear layout: both operands K-contiguous)
//
// v1 structure ("one barrier per K-tile"):
//   * 128x128 block tile, BK = 64, 256 threads = 4 waves in 2x2, each wave a 64x64 sub-tile = 4x4 MFMA 16x16x32 tiles
//   * both operand tiles are staged HBM/L2 -> LDS with global_load_lds dwordx4 (no VGPR round trip), double-buffered:
//     tile t+1 is issued before the MFMAs of tile t, one vmcnt(0)+barrier per K-tile
//   * LDS rows are 128 B with the 16-B slot XOR-swizzled by (row>>1)&7 (common.h) -> conflict-free ds_read_b128
//   * fused epilogues: bias / activation / LayerScale+residual / V-transpose scatter / patch-embed row remap + pos
//   * blockIdx -> tile mapping is XCD-aware (consecutive tiles share an XCD L2)
//
// MFMA operand roles.  __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c): lane l supplies a = Arow[l&15][8*(l>>4)..+8],
// b = Bcol[l&15][8*(l>>4)..+8]; result lane l holds D[row = 4*(l>>4)+r][col = l&15], r = 0..3.
//   SWAP = true  (default): a = W fragment (row = n), b = X fragment (col = m)  -> lane holds C[m = l&15][n = 4g+r]:
//                four consecutive n  -> one 8-byte bf16x4 store per fragment into row-major C.
//   SWAP = false (V^T epilogue): a = X fragment, b = W fragment -> lane holds C[m = 4g+r][n = l&15]:
//                four consecutive m -> one 8-byte store into the token-contiguous V^T layout.
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "gemm_epilogue.h"
#include "visrep_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;      // 16 KB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // double buffer = 64 KB

__device__ __attribute__((aligned(16))) uint32_t g_zero_page[4] = {0u, 0u, 0u, 0u};   // source of padded (out-of-image) conv taps

template <int EPI, bool CONV = false, bool GN = false>        // GN (with CONV): the epilogue also emits GroupNorm partial sums (GemmArgs::gn_partial)
__global__ __launch_bounds__(256, 2) void gemm_bf16_128(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + BN - 1) / BN;                      // N % 64 == 0: the last tile column may be half empty
    const int ntm = (p.M + BM - 1) / BM;
    const int t = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;

    // split-K: slice blockIdx.y covers K columns [y*kslice, (y+1)*kslice) and writes its own fp32 [M, ldc] plane
    const int kbase = p.kslice ? (int)blockIdx.y * p.kslice : 0;
    // ---- staging addresses: 16-B chunk c = j*256 + tid, row = c>>3 = j*32 + (tid>>3), physical slot = tid&7
    const int srow = tid >> 3;
    const int lslot = (tid & 7) ^ ((srow >> 1) & 7);          // logical slot this lane must fetch (swizzle on the source)
    const bf16_t* ga[4];
    const bf16_t* gw[4];
    int iy0[4], ix0[4], pb[4];                                 // CONV: top-left tap coordinates and image pixel base per staged row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ra = m0 + j * 32 + srow;
        ra = ra < p.M ? ra : p.M - 1;                          // clamp: rows past M are computed but never stored
        if (CONV) {
            const int hw = p.cHo * p.cWo;
            const int rc = ra + p.a_row0;                      // a tail launch of the dispatcher: its rows follow the head's
            const int b = rc / hw, pix = rc - b * hw;
            const int oy = pix / p.cWo, ox = pix - oy * p.cWo;
            iy0[j] = oy * p.cstride - p.cpad;
            ix0[j] = ox * p.cstride - p.cpad;
            pb[j] = b * p.cH * p.cW;
            ga[j] = p.A + lslot * 8;
        } else {
            ga[j] = p.A + (size_t)visrep_a_row(p, ra) * p.lda + kbase + lslot * 8;
        }
        int rw = n0 + j * 32 + srow;
        rw = rw < p.N ? rw : p.N - 1;
        gw[j] = p.W + (size_t)rw * p.ldw + kbase + lslot * 8;
    }
    // CONV: running (tap, channel offset) of the next K-tile to stage; stage() is called for kt = 0, 1, 2, ... in order
    int tap = 0, c0 = 0;
    if (CONV) { tap = kbase / p.cC; c0 = kbase - tap * p.cC; }
    const int Hl = p.cH << p.cup, Wl = p.cW << p.cup;          // logical (nearest-2x upsampled) source size
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE_BYTES + wave * 1024;
        char* sw = sa + TILE_BYTES;
        const int ko = kt * BK;
        if (CONV) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = iy0[j] + ky, ix = ix0[j] + kx;
                const bool ok = (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
                const bf16_t* src = ga[j] + ((size_t)(pb[j] + (iy >> p.cup) * p.cW + (ix >> p.cup)) * p.cC + c0);
                glds16(ok ? src : reinterpret_cast<const bf16_t*>(g_zero_page), sa + j * 4096);
            }
            c0 += BK;
            if (c0 == p.cC) { c0 = 0; ++tap; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(ga[j] + ko, sa + j * 4096);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(gw[j] + ko, sw + j * 4096);
    };

    // ---- fragment read offsets (bytes inside a tile); row = base16 + (lane&15), logical slot = kk*4 + (lane>>4)
    const int fr = lane & 15, fg = lane >> 4;
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = fr * 128 + (((kk * 4 + fg) ^ ((fr >> 1) & 7)) << 4);
    const int aoff = wm * 64 * 128, woff = TILE_BYTES + wn * 64 * 128;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.kslice ? p.kslice : p.K) / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xa[4], xw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(sb + aoff + i * 2048 + foff[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i) xw[i] = *reinterpret_cast<const bf16x8*>(sb + woff + i * 2048 + foff[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (EPI == EPI_VT)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i], xw[j], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw[j], xa[i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();   // waits the in-flight LDS-DMA of tile kt+1 (vmcnt(0)) and fences the reads of tile kt
    }

    // ------------------------------------------------------------------ epilogues (gemm_epilogue.h)
    if (n0 + wn * 64 >= p.N) return;                          // wave-uniform: the empty half of an N-edge tile
    if (EPI == EPI_F32 && p.kslice) {
        GemmArgs q = p;
        q.C = reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(p.C) + (size_t)blockIdx.y * p.M * p.ldc);
        gemm_epilogue_rowmajor<EPI, 4, 4>(q, acc, m0 + wm * 64, n0 + wn * 64, fr, fg);
        return;
    }
    if constexpr (GN) { gemm_epilogue_rowmajor_gn<EPI, 4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fg); return; }
    if (EPI == EPI_VT) gemm_epilogue_vt<4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fg);
    else gemm_epilogue_rowmajor<EPI, 4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fg);
}

// split-K reduction for the V^T epilogue: thread = (4 consecutive tokens m, one feature n); vt[n * ldc + perm16(m)]
__global__ __launch_bounds__(256) void splitk_reduce_vt(const float* __restrict__ part, int S, const GemmArgs p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int mq = (p.M + 3) >> 2;
    if (idx >= (long)mq * p.N) return;
    const int n = (int)(idx % p.N), m = (int)(idx / p.N) * 4;   // n fastest: coalesced reads of the fp32 planes
    const size_t plane = (size_t)p.M * p.N;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (m + e < p.M) {
            float a = part[(size_t)(m + e) * p.N + n];
            for (int s = 1; s < S; ++s) a += part[s * plane + (size_t)(m + e) * p.N + n];
            if (p.ln_rt) { const float2 rt = p.ln_rt[visrep_a_row(p, m + e)]; a = __builtin_fmaf(a, rt.x, rt.y * p.ln_s[n]); }
            v[e] = a + (p.bias ? p.bias[n] : 0.f);
        }
    }
    const int mp = (m & ~15) | ((((m >> 2) & 1) << 1 | ((m >> 3) & 1)) << 2);   // perm16: swap 4-token groups 1 <-> 2
    u32x2 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    *reinterpret_cast<u32x2*>(p.C + (size_t)n * p.ldc + mp) = o;
}

template <int EPI, bool CONV = false, bool GN = false>
int launch(const GemmArgs& a, hipStream_t s) {
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    static VisrepLdsOptIn opt;                                      // per (kernel instantiation, device)
    visrep_lds_opt_in(opt, reinterpret_cast<const void*>(gemm_bf16_128<EPI, CONV, GN>), LDS_BYTES);
    hipLaunchKernelGGL((gemm_bf16_128<EPI, CONV, GN>), dim3(ntm * ntn, a.kslice ? a.K / a.kslice : 1), dim3(256), LDS_BYTES, s, a);
    return hipGetLastError() == hipSuccess ? 0 : VISREP_ERR_LAUNCH;
}

// ---- split-K reduction: out = epilogue( sum_s part[s] ), slices added in index order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ part, int S, const GemmArgs p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // one thread per 4 consecutive n
    const int nv = p.N >> 2;
    if (idx >= (long)p.M * nv) return;
    const int m = (int)(idx / nv), n = (int)(idx - (long)m * nv) * 4;
    const size_t plane = (size_t)p.M * p.N;
    float4 acc = *reinterpret_cast<const float4*>(part + (size_t)m * p.N + n);
    for (int s = 1; s < S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(part + s * plane + (size_t)m * p.N + n);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (p.ln_rt) {                                             // LayerNorm folded into the GEMM: rstd * (acc - mean * s)
        const float2 rt = p.ln_rt[visrep_a_row(p, m)];
        const float4 sv = *reinterpret_cast<const float4*>(p.ln_s + n);
        acc.x = __builtin_fmaf(acc.x, rt.x, rt.y * sv.x); acc.y = __builtin_fmaf(acc.y, rt.x, rt.y * sv.y);
        acc.z = __builtin_fmaf(acc.z, rt.x, rt.y * sv.z); acc.w = __builtin_fmaf(acc.w, rt.x, rt.y * sv.w);
    }
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    if (p.epi == EPI_ACT) { acc.x = apply_act(acc.x, p.act); acc.y = apply_act(acc.y, p.act); acc.z = apply_act(acc.z, p.act); acc.w = apply_act(acc.w, p.act); }
    if (p.epi == EPI_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = acc;
        return;
    }
    if (p.epi == EPI_RESID) {
        if (p.ls) {
            const float4 l = *reinterpret_cast<const float4*>(p.ls + n);
            acc.x *= l.x; acc.y *= l.y; acc.z *= l.z; acc.w *= l.w;
        }
        const u32x2 r = *reinterpret_cast<const u32x2*>(p.resid + (size_t)m * p.ldc + n);
        acc.x += bf_lo(r[0]); acc.y += bf_hi(r[0]); acc.z += bf_lo(r[1]); acc.w += bf_hi(r[1]);
    }
    u32x2 o = {pack_bf16(acc.x, acc.y), pack_bf16(acc.z, acc.w)};
    *reinterpret_cast<u32x2*>(p.C + (size_t)m * p.ldc + n) = o;
    if (p.stat_rt && p.N == 1024) {
        // N = 4 * blockDim: this block IS row m, so the row's LayerNorm statistics (of the bf16-ROUNDED outputs, the formula of
        // ln_stats_finalize_rows) come out of the reduction itself instead of a read-back pass over the rows (splitk_reduce_emits_stats)
        __shared__ float red[2][4];
        const float r0 = bf_lo(o[0]), r1 = bf_hi(o[0]), r2 = bf_lo(o[1]), r3 = bf_hi(o[1]);
        const float s1 = wave_sum((r0 + r1) + (r2 + r3));
        const float s2 = wave_sum(__builtin_fmaf(r0, r0, __builtin_fmaf(r1, r1, __builtin_fmaf(r2, r2, r3 * r3))));
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), t2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
            const float mean = t1 * (1.0f / 1024.0f);
            const float var = fmaxf(t2 * (1.0f / 1024.0f) - mean * mean, 0.f);
            const float rstd = 1.0f / sqrtf(var + p.stat_eps);
            p.stat_rt[m] = float2{rstd, -mean * rstd};
        }
    }
}
bool splitk_reduce_emits_stats(const GemmArgs& a) { return a.stat_rt && a.epi == EPI_RESID && a.N == 1024; }

}  // namespace

thread_local int t_visrep_gemm_variant = 5;
thread_local int t_visrep_gemm_walk = 0;
extern "C" int visrep_set_gemm_walk(int code) {                  // per-thread A/B knob: tile order of the persistent 256x256 kernel; returns the previous value
    const int old = t_visrep_gemm_walk;
    t_visrep_gemm_walk = code < 0 ? 0 : code;
    return old;
}
int g_visrep_gemm_dbg = 0;
unsigned long long* g_visrep_gemm_dbg_buf = nullptr;

// ---- per-device state -------------------------------------------------------------------------------------------------------------
int visrep_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < VISREP_MAX_DEVICES ? dev : VISREP_MAX_DEVICES - 1;
}
// CUs the persistent kernels leave free (round 6): with world > 1 the C leg's all_to_all runs on RCCL's own kernels WHILE the next tower launch's
// persistent GEMMs would otherwise own every CU (sweep.c_score_of); reserving k CUs (a multiple of 8: one or more per XCD) gives the collective
// somewhere to run.  Process-wide, off by default (0): no multi-GPU box has been available to A/B it.  VISREP_RESERVE_CUS=k in the environment
// or visrep_set_reserved_cus(k); grids captured into a HIP graph keep the size they were captured with.
static std::atomic<int> g_reserved_cus{-1};
static int reserved_cus() {
    int v = g_reserved_cus.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("VISREP_RESERVE_CUS");
        v = e ? atoi(e) : 0;
        v = v < 0 ? 0 : v / 8 * 8;
        g_reserved_cus.store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" int visrep_set_reserved_cus(int k) {
    const int old = reserved_cus();
    g_reserved_cus.store(k < 0 ? 0 : k / 8 * 8, std::memory_order_relaxed);
    return old;
}
int visrep_cu_count() {
    static std::atomic<int> n[VISREP_MAX_DEVICES];
    const int dev = visrep_device();
    int v = n[dev].load(std::memory_order_relaxed);
    if (!v) {
        hipDeviceProp_t prop;
        v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        n[dev].store(v, std::memory_order_relaxed);
    }
    const int r = reserved_cus();
    return (r > 0 && v - r >= 64) ? v - r : v;                   // never below 64 CUs: a mistyped reservation must not serialise the chip
}

// split-K scratch registry: (device, stream) -> caller-owned buffer.  A handful of entries per device, searched under a mutex (a launch-path
// lookup is a few loads; registration happens once per engine).
namespace {
struct ScratchEntry { bool used, any; hipStream_t stream; void* ptr; size_t bytes; };
constexpr int SCRATCH_SLOTS = 9;                                 // 1 device-wide + 8 stream-keyed registrations per device
ScratchEntry g_scratch[VISREP_MAX_DEVICES][SCRATCH_SLOTS] = {};
std::mutex g_scratch_mu;
}  // namespace
VisrepScratch visrep_scratch_for(hipStream_t s) {
    const int dev = visrep_device();
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    VisrepScratch any{nullptr, 0};
    for (const ScratchEntry& e : g_scratch[dev]) {
        if (!e.used) continue;
        if (!e.any && e.stream == s) return VisrepScratch{e.ptr, e.bytes};
        if (e.any) any = VisrepScratch{e.ptr, e.bytes};
    }
    return any;
}
int visrep_scratch_register(bool any_stream, hipStream_t s, void* ptr, size_t bytes) {
    const int dev = visrep_device();
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ScratchEntry* slot = nullptr;
    for (ScratchEntry& e : g_scratch[dev])
        if (e.used && e.any == any_stream && (any_stream || e.stream == s)) slot = &e;
    if (!ptr) {                                                   // unregister
        if (slot) slot->used = false;
        return 0;
    }
    if (!slot)
        for (ScratchEntry& e : g_scratch[dev])
            if (!e.used) { slot = &e; break; }
    if (!slot) return visrep_set_error(VISREP_ERR_ARG, "set_stream_scratch: at most 8 stream-keyed registrations per device");
    *slot = ScratchEntry{true, any_stream, s, ptr, bytes};
    return 0;
}

namespace {
// variant 6: the duo kernel (gemm_bf16_duo.hip: two independent 4-wave workgroups per CU, 256 x 128 tiles) wherever it supports the problem;
// variant 7: for N <= 1024 only (out-proj, fc2, V^T: the shapes whose epilogue is the largest share of a tile).  Everything else as variant 5.
bool duo_wanted(const GemmArgs& a, int variant) {
    return variant >= 6 && visrep_gemm_duo_supports(a) && !a.gn_partial && a.a_period <= 0 && (variant != 7 || a.N <= 1024);
}
int dispatch_one(const GemmArgs& a, hipStream_t s, int variant) {
    if (duo_wanted(a, variant)) {
        visrep_count_route(VISREP_ROUTE_GEMM_256);
        return visrep_gemm_duo_dispatch(a, s, variant != 8);       // 8 = the unpipelined first build (kept for the table in profiles/round6_gemm.md)
    }
    if (variant >= 6) variant = 5;
    if (a.conv) {                                              // implicit 3x3 convolution: the 256x256 kernel when whole rounds of its tiles exist
#ifndef VISREP_NO_CONV5                                           // A/B builds (tools/): every convolution on the 128x128 kernel, as until round 4
        if (variant == 5 && visrep_gemm_v5_supports_conv(a) && (long)((a.M + 255) / 256) * (a.N / 256) >= 2L * visrep_cu_count()) {
            visrep_count_route(VISREP_ROUTE_CONV_256);
            return visrep_gemm_v5_dispatch(a, s);
        }
#endif
        visrep_count_route(a.gn_partial ? VISREP_ROUTE_CONV_128_GN : VISREP_ROUTE_CONV_128);
        if (a.gn_partial) return a.epi == EPI_BIAS ? launch<EPI_BIAS, true, true>(a, s) : launch<EPI_RESID, true, true>(a, s);
        switch (a.epi) {
            case EPI_BIAS: return launch<EPI_BIAS, true>(a, s);
            case EPI_RESID: return launch<EPI_RESID, true>(a, s);
            case EPI_F32: return launch<EPI_F32, true>(a, s);
        }
        return visrep_set_error(VISREP_ERR_ARG, "conv3x3: epilogue must be BIAS, RESID or F32");
    }
#ifdef VISREP_EXPERIMENTS                                         // v3 / v4: documented dead ends, tools-only library (build.build_experiments_lib)
    if (variant == 4 && visrep_gemm_v4_supports(a)) return visrep_gemm_v4_dispatch(a, s);
#endif
    if (variant == 5 && visrep_gemm_v5_supports(a)) {
        GemmArgs b = a;
        b.dbg = g_visrep_gemm_dbg;
        b.dbg_buf = g_visrep_gemm_dbg_buf;
        visrep_count_route(VISREP_ROUTE_GEMM_256);
        return visrep_gemm_v5_dispatch(b, s);
    }
#ifdef VISREP_EXPERIMENTS
    if (variant == 3 && visrep_gemm_v3_supports(a)) {
        GemmArgs b = a;
        b.dbg = g_visrep_gemm_dbg;
        return visrep_gemm_v3_dispatch(b, s);
    }
#endif
    if (variant >= 2 && visrep_gemm_v2_supports(a)) {
        GemmArgs b = a;
        b.dbg = g_visrep_gemm_dbg;
        b.dbg_buf = g_visrep_gemm_dbg_buf;
        visrep_count_route(VISREP_ROUTE_GEMM_256);
        return visrep_gemm_v2_dispatch(b, s);
    }
    visrep_count_route(VISREP_ROUTE_GEMM_128);
    switch (a.epi) {
        case EPI_BIAS: return launch<EPI_BIAS>(a, s);
        case EPI_ACT: return launch<EPI_ACT>(a, s);
        case EPI_RESID: return launch<EPI_RESID>(a, s);
        case EPI_VT: return launch<EPI_VT>(a, s);
        case EPI_PATCH: return launch<EPI_PATCH>(a, s);
        case EPI_F32: return launch<EPI_F32>(a, s);
    }
    return visrep_set_error(VISREP_ERR_ARG, "gemm: unknown epilogue");
}

int cu_count() { return visrep_cu_count(); }
}  // namespace

namespace {
// Few output tiles but a deep reduction (3x3 convolutions of the diffusion towers at 12x12 / 24x24 resolution: M = 144 .. 576,
// K = 9 * 1280 .. 9 * 2560; the 128x128 tail launches of the ViT GEMMs): a handful of CUs would walk K serially.  Split K over
// blockIdx.y into fp32 planes in the caller's scratch (visrep_set_scratch) and reduce them in slice order with the epilogue
// fused.  Returns 1 when the problem was handled this way, 0 when it was not eligible, < 0 on error.
int try_split_k(const GemmArgs& a, hipStream_t s) {
    if (!(a.epi != EPI_PATCH && a.K >= 1024 && (a.N & 3) == 0) || a.gn_partial) return 0;     // GroupNorm partials come from a GEMM epilogue only
    const VisrepScratch reg = visrep_scratch_for(s);               // the buffer of (this device, this stream), else the device-wide one
    void* const scratch = reg.ptr;
    const size_t scratch_bytes = reg.bytes;
    if (!scratch) return 0;
    const int ncu = cu_count();
    const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    // the 128x128 kernel keeps two blocks per CU resident: 2 * ncu block slots; fill them when the tiles alone do not
    if (tiles >= 2L * ncu) return 0;
    const int kt = a.K / BK;
    int S = (int)(2L * ncu / tiles);
    if (S > kt / 4) S = kt / 4;                                   // at least 4 K-tiles per slice
    while (S > 1 && kt % S) --S;
    if (S <= 1 || (size_t)S * a.M * a.N * sizeof(float) > scratch_bytes) return 0;
    GemmArgs part = a;
    part.C = reinterpret_cast<bf16_t*>(scratch);
    part.ldc = a.N;
    part.epi = EPI_F32; part.bias = nullptr; part.resid = nullptr; part.ls = nullptr; part.ln_rt = nullptr; part.ln_s = nullptr;
    part.kslice = a.K / S;
    visrep_count_route(VISREP_ROUTE_SPLITK);
    const int rc = a.conv ? launch<EPI_F32, true>(part, s) : launch<EPI_F32>(part, s);
    if (rc) return rc;
    if (a.epi == EPI_VT) {
        const long nthreads = (long)((a.M + 3) / 4) * a.N;
        hipLaunchKernelGGL(splitk_reduce_vt, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, (const float*)scratch, S, a);
    } else {
        const long nthreads = (long)a.M * (a.N / 4);
        hipLaunchKernelGGL(splitk_reduce, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, (const float*)scratch, S, a);
    }
    return hipGetLastError() == hipSuccess ? 1 : visrep_set_error(VISREP_ERR_LAUNCH, "gemm: split-K reduce launch failed");
}
}  // namespace

namespace {
// the statistics-emitting epilogue exists in the 256x256 v2 kernel only
bool v2_emits_stats(const GemmArgs& a, int variant) {
    return a.stat_rt && a.stat_partial && a.epi == EPI_RESID && !a.conv && (variant == 2 || variant >= 5) && visrep_gemm_v2_supports(a) && a.N % 128 == 0;   // v5 falls back to v2 when K % 64 != 0: both emit
}
// rows [0, a.M) of a finished residual GEMM -> a.stat_rt: from the epilogue's partial sums, else by reading the rows back
int finish_stats(const GemmArgs& a, bool from_partials, hipStream_t s) {
    if (!a.stat_rt) return 0;
    if (from_partials) return visrep_ln_stats_finalize(a.stat_partial, a.N / 64, a.stat_rt, a.M, a.N, a.stat_eps, s);
    return visrep_layernorm_stats(a.C, a.ldc, a.stat_rt, a.M, a.N, a.stat_eps, s);
}
int run_one(const GemmArgs& a, hipStream_t s, int variant) {
    GemmArgs b = a;
    const bool emit = v2_emits_stats(a, variant);
    if (emit) b.stat_slots = a.N / 64; else b.stat_partial = nullptr;
    const int rc = dispatch_one(b, s, variant);
    if (rc == VISREP_ERR_LAUNCH) return visrep_set_error(VISREP_ERR_LAUNCH, "gemm: launch failed");   // appends a failed LDS opt-in's cause
    return rc ? rc : finish_stats(a, emit, s);
}
int run_split_k(const GemmArgs& a, hipStream_t s) {              // 1 = handled, 0 = not eligible, < 0 = error
    const int sk = try_split_k(a, s);
    if (sk <= 0) return sk;
    if (splitk_reduce_emits_stats(a)) return 1;                  // the reduction wrote stat_rt for its rows
    const int rc = finish_stats(a, false, s);
    return rc ? rc : 1;
}
}  // namespace

namespace {
bool conv5_ok(const GemmArgs& a, int variant) {
#ifdef VISREP_NO_CONV5
    return false;
#else
    return variant >= 5 && visrep_gemm_v5_supports_conv(a);
#endif
}
}  // namespace

int visrep_gemm_dispatch(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "gemm: empty problem");
    if (a.N % 64 != 0 || a.K % BK != 0) return visrep_set_error(VISREP_ERR_SHAPE, "gemm: N and K must be multiples of 64");
    if ((a.lda % 8) || (a.ldw % 8) || (a.ldc % 4)) return visrep_set_error(VISREP_ERR_SHAPE, "gemm: leading dimensions must keep 16-B row alignment");
    if (a.stat_rt && a.epi != EPI_RESID) return visrep_set_error(VISREP_ERR_ARG, "gemm: row statistics are an EPI_RESID feature");
    if (a.a_period > 0 && (a.conv || a.epi == EPI_RESID || a.epi == EPI_PATCH || a.a_stride < a.a_period || a.a_first < 0))
        return visrep_set_error(VISREP_ERR_ARG, "gemm: the A row map serves plain BIAS / ACT / VT / F32 epilogues only");
    if (a.gn_partial && (!a.conv || (a.epi != EPI_BIAS && a.epi != EPI_RESID) || (a.gn_cpg != 4 && a.gn_cpg != 8 && a.gn_cpg != 16) || a.gn_hw <= 0 ||
                         a.gn_hw % 128 || a.M % a.gn_hw || a.N % a.gn_cpg || a.N % 64))
        return visrep_set_error(VISREP_ERR_ARG, "gemm: GroupNorm partials need a convolution, bias / residual epilogue, 4 | 8 | 16 channels per group, HW % 128 == 0");
    const int variant = t_visrep_gemm_variant;
    {
        const int sk = run_split_k(a, s);
        if (sk) return sk < 0 ? sk : 0;
    }
    // Tile quantisation: the persistent 256x256 kernels run one block per CU, so T tiles cost ceil(T / CUs) tile-times.
    // The BASELINE shapes have M = 256 * 577 (577 is prime): 2308 / 4616 / 9232 tiles = 9 / 18 / 36 full rounds + a 4..16-tile
    // remainder that would cost a whole extra round on 252 idle CUs.  When the remainder is small, the rows of the last
    // round are split off and run as 128x128 tiles (v1), which spread over many CUs and finish in a fraction of a round.
    if (variant >= 2 && a.N % 256 == 0 && a.epi != EPI_PATCH && (!a.conv || conv5_ok(a, variant))) {
        const bool duo = duo_wanted(a, variant);                         // 256 x 128 tiles, two workgroups per CU: rounds of 2 x CUs tiles
        const int ncu = duo ? 2 * cu_count() : cu_count(), ntn = a.N / (duo ? 128 : 256), ntm = (a.M + 255) / 256;
        const long tiles = (long)ntm * ntn, rounds = tiles / ncu, rem = tiles % ncu;
        if (rounds >= 1 && rem > 0 && rem * 4 <= ncu && (rounds * ncu) % ntn == 0) {
            const int m1 = (int)(rounds * ncu / ntn) * 256;             // rows covered by the full rounds
            if (m1 > 0 && m1 < a.M) {
                GemmArgs head = a, tail = a;
                head.M = m1;
                tail.M = a.M - m1;
                if (a.a_period > 0 || a.conv) tail.a_row0 = a.a_row0 + m1;   // row-mapped A / convolution: the map carries the offset, the pointers stay
                else tail.A = a.A + (size_t)m1 * a.lda;
                if (a.epi == EPI_VT) tail.C = a.C + m1;                  // V^T: token axis is the column axis (m1 % 16 == 0 keeps perm16)
                else if (a.epi == EPI_F32) tail.C = reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(a.C) + (size_t)m1 * a.ldc);
                else tail.C = a.C + (size_t)m1 * a.ldc;
                if (a.resid) tail.resid = a.resid + (size_t)m1 * a.ldc;
                if (a.ln_rt && a.a_period <= 0) tail.ln_rt = a.ln_rt + m1;
                if (a.stat_rt) tail.stat_rt = a.stat_rt + m1;
                const int rc = run_one(head, s, variant);
                if (rc) return rc;
                visrep_count_route(VISREP_ROUTE_GEMM_TAIL);
                const int sk = run_split_k(tail, s);                     // the tail has few tiles: split its K loop when it pays
                return sk ? (sk < 0 ? sk : 0) : run_one(tail, s, 1);
            }
        }
    }
    return run_one(a, s, variant);
}
