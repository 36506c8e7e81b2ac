// Internal (non-ABI) declarations shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

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
    // Row map of the A operand (and of ln_rt, which is indexed like A's rows), a_period > 0: logical row r of this GEMM is the physical row
    //   ((r + a_row0) / a_period) * a_stride + (r + a_row0) % a_period + a_first
    // of A - "rows 1 .. T-1 of every image" (a_period = T - 1, a_stride = T, a_first = 1: the V projection over the patch tokens, written into
    // image-aligned V^T columns) or "row 0 of every image" (a_period = 1, a_stride = T: the CLS tokens).  a_row0 carries the logical offset of
    // a tail launch (the dispatcher does not shift A / ln_rt then).  0 = identity.
    int a_period, a_stride, a_first, a_row0;
    // implicit 3x3 convolution (v1 kernel, conv = 1): A is the channels-last activation [B, cH, cW, cC] and the A tile of
    // K-tile (tap, c0) is gathered on the fly: row m = (b, oy, ox) reads x[b, (oy*cstride+ky-cpad)>>cup, (ox*cstride+kx-cpad)>>cup, c0..]
    int conv, cH, cW, cC, cHo, cWo, cstride, cpad, cup;
    // Convolution that also emits the GroupNorm statistics of its OUTPUT (the next ResnetBlock2D norm): per 64-row slot of an image and
    // per group the sum and the sum of squares of the (fp32, unrounded) outputs -> gn_partial[((b * nblk + slot) * G + g)], nblk = gn_hw / 64,
    // G = N / gn_cpg; groupnorm_finalize adds a (image, group)'s slots in order.  Needs gn_hw % 128 == 0 (a wave's 64 / 128 rows lie inside one
    // image), gn_cpg in {4, 8, 16} (a group = one / two / four lanes' column quads of a 16-column block), EPI_BIAS | EPI_RESID, no split-K.
    float2* gn_partial;
    int gn_cpg, gn_hw;
    // EPI_F32X (variant 5 only): an fp32 GEMM on the bf16 matrix pipe.  A and W hold the bf16 PLANES of fp32 matrices side by side
    // (hi | mid [| lo]: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); two planes carry 16 significand bits, three carry 24):
    // A [M, nplanes * ksplit], W [N, nplanes * ksplit].  K = nprod * ksplit walks nprod (A plane, W plane) pairs, two bits per pair in
    // tab_a / tab_w (split_tables()):  6 products = every pair whose terms are >= 2^-24 of the result - (hi,hi) (hi,mid) (hi,lo) (mid,hi)
    // (mid,mid) (lo,hi), fp32-equivalent;  4 = the full product of two-plane operands;  3 = (hi,hi) (hi,mid) (mid,hi), terms < 2^-16 dropped.
    // Epilogue in fp32: v = act(acc + bias); v = resid32 + ls * v (if resid32); C (fp32, may be null) and / or the planes of v ->
    // planes [M, out_planes * N] (leading dimension ldp).
    int ksplit;
    unsigned tab_a, tab_w;
    const float* resid32;
    bf16_t* planes;
    int ldp, out_planes;
    // XCD-weighted tile split of the persistent 256x256 kernel (gemm_bf16_v5.hip, visrep_xcd_plan): the eight XCDs of an MI355X run at
    // their own clocks under the power limit (measured: +-4 % around the mean inside one launch), so with equal shares of the tile list the
    // launch ends when the slowest XCD does.  xcd_bounds[x] .. xcd_bounds[x + 1] = the tile indices of XCD x (xcd_bounds[8] = 0: equal
    // shares); xb / xb_host / xb_seq: a sampled launch (xb != null) records when each XCD finished, its last block turns that into tile times
    // and leaves them in pinned host memory, from where the next plans are made.  Results do not depend on the split.
    struct VisrepXcdDev* xb;
    struct VisrepXcdHost* xb_host;
    unsigned xb_seq;
    int xcd_bounds[9];
    int walk;                       // v5 tile walk (set by the launcher from t_visrep_gemm_walk): 0 = default; low byte C > 0 = column-group-major walk, C columns per group
    unsigned long long* dbg_buf;    // timing-only: per-segment cycle sums (VISREP_GEMM_ABLATE builds)
    int dbg;                        // timing-only ablation mask for the v2 kernel (1 = no MFMA, 2 = no LDS-DMA, 4 = no ds_read); 0 in production
};

struct VisrepXcdSlot { unsigned long long t_end[8]; unsigned long long t_start; unsigned done, pad; };   // 100-MHz ticks (s_memrealtime)
struct VisrepXcdDev { VisrepXcdSlot slot[16]; };                      // device memory; a slot serves one sampled launch at a time and is left zeroed by its last block
struct VisrepXcdHost { float tile_ticks[8]; unsigned seq; unsigned pad; };   // pinned host memory: the last sampled launch's time per round of tiles, per XCD

#ifdef __HIPCC__
__device__ __forceinline__ int visrep_a_row(const GemmArgs& p, int r) {          // logical -> physical row of A / ln_rt
    if (p.a_period <= 0) return r;
    const int g = r + p.a_row0, q = g / p.a_period;
    return q * p.a_stride + (g - q * p.a_period) + p.a_first;
}
#endif
int visrep_gemm_dispatch(const GemmArgs& a, hipStream_t s);
int visrep_ln_stats_finalize(const float2* partial, int slots, float2* rt, int rows, int d, float eps, hipStream_t s);
bool visrep_gemm_v2_supports(const GemmArgs& a);
int visrep_gemm_v2_dispatch(const GemmArgs& a, hipStream_t s);
bool visrep_gemm_v4_supports(const GemmArgs& a);
int visrep_gemm_v4_dispatch(const GemmArgs& a, hipStream_t s);
bool visrep_gemm_v5_supports(const GemmArgs& a);
bool visrep_gemm_duo_supports(const GemmArgs& a);                 // gemm_bf16_duo.hip (variants 6 / 7: A/B only)
int visrep_gemm_duo_dispatch(const GemmArgs& a, hipStream_t s, bool pipelined);
// Fills a.xcd_bounds (and, on every 8th launch of this (kernel, M, N, K), a.xb / a.xb_host / a.xb_seq) for a launch of `grid` persistent blocks
// over `ntiles` output tiles on the current device.  VISREP_XCD_BALANCE=0 / visrep_set_xcd_balance(0): equal shares, nothing recorded.
void visrep_xcd_plan(GemmArgs& a, hipStream_t s, int grid, int ntiles, const void* kernel);
bool visrep_gemm_v5_supports_conv(const GemmArgs& a);
int visrep_gemm_v5_dispatch(const GemmArgs& a, hipStream_t s);
bool visrep_gemm_v3_supports(const GemmArgs& a);
int visrep_gemm_v3_dispatch(const GemmArgs& a, hipStream_t s);
extern int g_visrep_gemm_dbg;
extern unsigned long long* g_visrep_gemm_dbg_buf;
// Kernel-variant selection is PER-THREAD state (visrep_set_*_variant changes the calling thread's choice only): two threads - or two
// engines on two GPUs driven from two threads - never see each other's diagnostic setting, and the defaults need no setter at all.
extern thread_local int t_visrep_gemm_variant;   // 1 = 128x128 kernel, 2 / 5 = 256x256 persistent ping-pong kernels (when N % 256 == 0); 3 / 4: VISREP_EXPERIMENTS builds; 6 / 7 = the duo kernel (two 4-wave workgroups per CU) everywhere it applies / for N <= 1024 only, 8 = its unpipelined first build; default 5
int visrep_attention_ab_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* out, int ldo,
                               int B, int Tq, int Tk, int H, int kv_shared, int causal, float scale, hipStream_t st);
extern thread_local int t_visrep_attn_variant;
int visrep_set_error(int code, const char* msg);
// Which kernel family a call was routed to - per-THREAD launch counters read by visrep_debug_routes() (tests assert that a shape takes the
// route it was tuned for; two integer adds per launch, no device work).  Indices are VISREP_ROUTE_* of include/visrep.h.
extern thread_local long t_visrep_routes[VISREP_ROUTE_COUNT];
inline void visrep_count_route(int r) { ++t_visrep_routes[r]; }
// split-bf16 product sets (GemmArgs::tab_a / tab_w, attn_f32_split_kernel): products in {3, 4, 6}; planes needed = 2, 2, 3
inline int visrep_split_planes(int products) { return products == 6 ? 3 : 2; }
inline bool visrep_split_tables(int products, unsigned& tab_a, unsigned& tab_w) {
    switch (products) {
        case 3: tab_a = 0x010u; tab_w = 0x004u; return true;      // A planes 0 0 1,       W planes 0 1 0
        case 4: tab_a = 0x050u; tab_w = 0x044u; return true;      // A planes 0 0 1 1,     W planes 0 1 0 1
        case 6: tab_a = 0x940u; tab_w = 0x124u; return true;      // A planes 0 0 0 1 1 2, W planes 0 1 2 0 1 0
    }
    return false;
}
constexpr int VISREP_MAX_DEVICES = 16;

// ---- per-DEVICE one-shot state: a process may drive several GPUs, and both the multiprocessor count and the function attribute
// hipFuncAttributeMaxDynamicSharedMemorySize belong to a device, not to the process
int visrep_device();      // current device index, clamped to [0, VISREP_MAX_DEVICES)
extern thread_local int t_visrep_gemm_walk;    // A/B knob (visrep_set_gemm_walk): tile order of the persistent 256x256 kernel
int visrep_cu_count();    // multiprocessors of the current device (cached per device)
struct VisrepLdsOptIn { std::atomic<int> bytes[VISREP_MAX_DEVICES]; };   // largest dynamic-LDS size opted in so far, per device (static storage: zeros)
// Raises the kernel's dynamic-LDS limit on the current device to at least `bytes`.  Returns hipSuccess or the error of hipFuncSetAttribute
// (launchers report "LDS opt-in failed" instead of a generic launch failure).  Race-free: set + record happen under one process-wide mutex
// (taken on the slow path only - a kernel's first launch per device and size), so two threads opting one kernel in at different sizes
// always leave the attribute at the larger value, which is also the value recorded.
hipError_t visrep_lds_opt_in_slow(VisrepLdsOptIn& st, const void* kernel, int bytes, int dev);
inline hipError_t visrep_lds_opt_in(VisrepLdsOptIn& st, const void* kernel, int bytes) {
    const int dev = visrep_device();
    if (st.bytes[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;
    return visrep_lds_opt_in_slow(st, kernel, bytes, dev);
}
// Caller-owned split-K scratch (visrep_set_scratch / visrep_set_stream_scratch): keyed by (device, stream).  A stream-keyed registration wins;
// the device-wide registration (stream key = "any") serves every other stream of that device - callers that run split-K GEMMs on several
// streams of one device concurrently register one buffer per stream.
struct VisrepScratch { void* ptr; size_t bytes; };
VisrepScratch visrep_scratch_for(hipStream_t s);
int visrep_scratch_register(bool any_stream, hipStream_t s, void* ptr, size_t bytes);
