// C-ABI glue: error handling, the standalone GEMM entry point and the composed ViT tower forward.
#include <stdio.h>
#include <string.h>

#include <mutex>

#include <cstdlib>
#include "common.h"
#include "visrep_internal.h"

static thread_local char g_err[256] = "";
// A failed dynamic-LDS opt-in (visrep_lds_opt_in) leaves its HIP error here; the launch that follows fails too, and whatever message that
// launcher reports gets the real cause appended instead of a bare "launch failed".
static thread_local char g_lds_note[96] = "";

int visrep_set_error(int code, const char* msg) {
    if (code == VISREP_ERR_LAUNCH && g_lds_note[0]) {
        snprintf(g_err, sizeof(g_err), "%s (%s)", msg, g_lds_note);
        g_lds_note[0] = 0;
        return code;
    }
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return code;
}

hipError_t visrep_lds_opt_in_slow(VisrepLdsOptIn& st, const void* kernel, int bytes, int dev) {
    static std::mutex mu;                                        // slow path only: a kernel's first launch per (device, size)
    std::lock_guard<std::mutex> lock(mu);
    if (st.bytes[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;       // another thread raised it meanwhile
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        snprintf(g_lds_note, sizeof(g_lds_note), "LDS opt-in to %d bytes failed: %s", bytes, hipGetErrorString(e));
        (void)hipGetLastError();                                 // the sticky error belongs to this call, not to the launch behind it
        return e;
    }
    st.bytes[dev].store(bytes, std::memory_order_release);       // recorded only after the attribute holds: never larger than what is set
    g_lds_note[0] = 0;                                           // a stale note of an earlier failed opt-in must not ride on an unrelated later error
    return hipSuccess;
}

extern "C" int visrep_version(void) { return VISREP_VERSION; }

thread_local long t_visrep_routes[VISREP_ROUTE_COUNT] = {};

extern "C" int visrep_debug_routes(long* out, int reset) {
    for (int i = 0; i < VISREP_ROUTE_COUNT; ++i) {
        if (out) out[i] = t_visrep_routes[i];
        if (reset) t_visrep_routes[i] = 0;
    }
    return VISREP_ROUTE_COUNT;
}

namespace {
// visrep_debug_mfma_probe: a free-running MFMA stream (tools/probes/mfma_probe.hip, variant V0 / 16x16x32), operands in registers
__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(float* sink, int iters, int rnd) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    unsigned st = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ 0x9e3779b9u;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            if (rnd) {      // sign | biased exponent 0x7d..0x80 | random mantissa: |x| in [0.25, 4) - activations-sized values, every bit toggling
                st = st * 1664525u + 1013904223u; unsigned r = st >> 8;
                a[i][e] = (short)(((r & 1) << 15) | ((0x7d + ((r >> 1) & 3)) << 7) | ((r >> 3) & 0x7f));
                st = st * 1664525u + 1013904223u; r = st >> 8;
                b[i][e] = (short)(((r & 1) << 15) | ((0x7d + ((r >> 1) & 3)) << 7) | ((r >> 3) & 0x7f));
            } else {
                a[i][e] = (short)(e == 0 ? lane + i : e);
                b[i][e] = (short)(e == 0 ? lane * 3 + i : e);
            }
        }
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();   // shader-clock cycles / 100-MHz ticks
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        if ((it & 63) == 63) {                                   // keep the sums finite and the accumulator bits moving: restart from a small value
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] *= 1e-3f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][1];
    if (s == 123.456f) sink[0] = s;                              // keeps the accumulators live; never true in practice
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) {                   // the clock this stream ran at: (c1 - c0) / ((r1 - r0) * 10 ns)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(sink);
        o[1] = c1 - c0;
        o[2] = r1 - r0;
    }
}
}  // namespace

extern "C" int visrep_debug_mfma_probe(int iters, int random, void* sink, double* flop, void* stream) {
    if (iters <= 0 || !sink) return visrep_set_error(VISREP_ERR_ARG, "mfma_probe: iters > 0 and a device sink are required");
    const int ncu = visrep_cu_count();
    if (flop) *flop = (double)ncu * 8.0 * iters * 32.0 * 16384.0;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(ncu), dim3(512), 0, (hipStream_t)stream, (float*)sink, iters, random);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "mfma_probe: launch failed");
}

extern "C" int visrep_set_gemm_variant(int variant) {            // per-thread (see visrep_internal.h); returns the previous value
#ifdef VISREP_EXPERIMENTS
    const bool ok = variant >= 1 && variant <= 8;
#else
    const bool ok = variant == 1 || variant == 2 || variant == 5 || variant == 6 || variant == 7 || variant == 8;
#endif
    if (!ok) return visrep_set_error(VISREP_ERR_ARG, "gemm variant must be 1, 2, 5, 6, 7 or 8 (3 / 4: VISREP_EXPERIMENTS builds only)");
    const int old = t_visrep_gemm_variant;
    t_visrep_gemm_variant = variant;
    return old;
}

extern "C" int visrep_debug_gemm_ablation(int mask) {   // timing experiments only: results are WRONG when mask != 0
    const int old = g_visrep_gemm_dbg;
    g_visrep_gemm_dbg = mask;
    return old;
}

extern "C" int visrep_debug_gemm_timing_buffer(void* dev_u64x16) {   // ablation builds only: 16 x u64 per-segment cycle sums
    g_visrep_gemm_dbg_buf = (unsigned long long*)dev_u64x16;
    return 0;
}

extern "C" size_t visrep_last_error(char* buf, size_t n) {
    const size_t len = strlen(g_err);
    if (buf && n) {
        const size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return len;
}

extern "C" int visrep_device_cu_count(void) { return visrep_cu_count(); }

extern "C" int visrep_set_scratch(void* ptr, size_t bytes) {
    if ((ptr == nullptr) != (bytes == 0)) return visrep_set_error(VISREP_ERR_ARG, "set_scratch: pass (ptr, bytes) or (NULL, 0)");
    if ((size_t)ptr & 15) return visrep_set_error(VISREP_ERR_ARG, "set_scratch: pointer must be 16-byte aligned");
    return visrep_scratch_register(true, nullptr, ptr, bytes);   // device-wide registration of the CURRENT device
}

extern "C" int visrep_set_stream_scratch(void* stream, void* ptr, size_t bytes) {
    if ((ptr == nullptr) != (bytes == 0)) return visrep_set_error(VISREP_ERR_ARG, "set_stream_scratch: pass (ptr, bytes) or (NULL, 0)");
    if ((size_t)ptr & 15) return visrep_set_error(VISREP_ERR_ARG, "set_stream_scratch: pointer must be 16-byte aligned");
    return visrep_scratch_register(false, (hipStream_t)stream, ptr, bytes);   // keyed by (current device, stream)
}

extern "C" int visrep_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                                int K, int epilogue, int act, const void* resid, const float* ls, void* stream) {
    if (!A || !W || !C) return visrep_set_error(VISREP_ERR_ARG, "gemm: null pointer");
    if (epilogue == VISREP_EPI_PATCH) return visrep_set_error(VISREP_ERR_ARG, "gemm: EPI_PATCH is internal to visrep_vit_forward");
    if (epilogue == VISREP_EPI_RESID && !resid) return visrep_set_error(VISREP_ERR_ARG, "gemm: EPI_RESID needs resid");
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias;
    a.resid = (const bf16_t*)resid; a.ls = ls; a.pos = nullptr;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.epi = epilogue; a.act = act;
    return visrep_gemm_dispatch(a, (hipStream_t)stream);
}

extern "C" int visrep_gemm_bf16_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const void* ln_rt, const float* ln_s,
                                   void* C, int ldc, int M, int N, int K, int epilogue, int act, void* stream) {
    if (!A || !W || !C || !ln_rt || !ln_s) return visrep_set_error(VISREP_ERR_ARG, "gemm_ln: null pointer");
    if (epilogue != VISREP_EPI_BIAS && epilogue != VISREP_EPI_ACT && epilogue != VISREP_EPI_VT)
        return visrep_set_error(VISREP_ERR_ARG, "gemm_ln: epilogue must be BIAS, ACT or VT");
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias; a.ln_rt = (const float2*)ln_rt; a.ln_s = ln_s;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.epi = epilogue; a.act = act;
    return visrep_gemm_dispatch(a, (hipStream_t)stream);
}

extern "C" int visrep_gemm_bf16_rows(const void* A, int lda, int row_period, int row_stride, int row_first, const void* W, int ldw,
                                     const float* bias, const void* ln_rt, const float* ln_s, void* C, int ldc, int M, int N, int K, int epilogue,
                                     int act, void* stream) {
    if (!A || !W || !C || ((ln_rt == nullptr) != (ln_s == nullptr))) return visrep_set_error(VISREP_ERR_ARG, "gemm_rows: null pointer");
    if (row_period <= 0 || row_stride < row_period || row_first < 0) return visrep_set_error(VISREP_ERR_ARG, "gemm_rows: bad row map");
    if (epilogue == VISREP_EPI_VT && row_period % 4) return visrep_set_error(VISREP_ERR_ARG, "gemm_rows: EPI_VT needs row_period % 4 == 0");
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias; a.ln_rt = (const float2*)ln_rt; a.ln_s = ln_s;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.epi = epilogue; a.act = act;
    a.a_period = row_period; a.a_stride = row_stride; a.a_first = row_first;
    return visrep_gemm_dispatch(a, (hipStream_t)stream);
}

extern "C" int visrep_gemm_bf16_resid_stats(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                                            int K, const void* resid, const float* ls, void* rt, void* partial, float eps, void* stream) {
    if (!A || !W || !C || !resid || !rt || !partial) return visrep_set_error(VISREP_ERR_ARG, "gemm_resid_stats: null pointer");
    if (N > 2048) return visrep_set_error(VISREP_ERR_SHAPE, "gemm_resid_stats: N <= 2048");
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias; a.resid = (const bf16_t*)resid; a.ls = ls;
    a.stat_rt = (float2*)rt; a.stat_partial = (float2*)partial; a.stat_eps = eps;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.epi = EPI_RESID;
    return visrep_gemm_dispatch(a, (hipStream_t)stream);
}

extern "C" int visrep_conv3x3_bf16(const void* x, int B, int H, int W, int C, const void* Wt, int ldw, const float* bias, void* out, int ldc,
                                   int Cout, int stride, int pad_mode, int upsample, int epilogue, const void* resid, void* stream) {
    if (!x || !Wt || !out) return visrep_set_error(VISREP_ERR_ARG, "conv3x3: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3: empty problem");
    if (C % 64) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3: C must be a multiple of 64 (use visrep_im2col3x3 + visrep_gemm_bf16 otherwise)");
    if ((stride != 1 && stride != 2) || (pad_mode != 0 && pad_mode != 1) || (upsample != 0 && upsample != 1))
        return visrep_set_error(VISREP_ERR_ARG, "conv3x3: stride 1|2, pad_mode 0 (symmetric 1) | 1 (0,1,0,1), upsample 0|1");
    if (epilogue != VISREP_EPI_BIAS && epilogue != VISREP_EPI_RESID && epilogue != VISREP_EPI_F32)
        return visrep_set_error(VISREP_ERR_ARG, "conv3x3: epilogue must be BIAS, RESID or F32");
    if (epilogue == VISREP_EPI_RESID && !resid) return visrep_set_error(VISREP_ERR_ARG, "conv3x3: EPI_RESID needs resid");
    const int Hl = H << upsample, Wl = W << upsample, pad_total = pad_mode == 0 ? 2 : 1;
    GemmArgs a{};
    a.A = (const bf16_t*)x; a.W = (const bf16_t*)Wt; a.C = (bf16_t*)out; a.bias = bias; a.resid = (const bf16_t*)resid;
    a.conv = 1; a.cH = H; a.cW = W; a.cC = C; a.cstride = stride; a.cpad = pad_mode == 0 ? 1 : 0; a.cup = upsample;
    a.cHo = (Hl + pad_total - 3) / stride + 1;
    a.cWo = (Wl + pad_total - 3) / stride + 1;
    a.M = B * a.cHo * a.cWo; a.N = Cout; a.K = 9 * C; a.lda = 8; a.ldw = ldw; a.ldc = ldc; a.epi = epilogue;
    return visrep_gemm_dispatch(a, (hipStream_t)stream);
}

// Rows and groups must fit the epilogue's slots (64-row slots inside one image, a group inside a lane quad pair / quartet), and asking for the
// sums must not cost the convolution its kernel: the 128x128 kernel emits them from its epilogue for both epilogues, the 256x256 kernel (whole
// rounds of its tiles, Cout % 256 == 0) for EPI_BIAS (round 5: a pass over the accumulators in front of the epilogue); a residual
// convolution there keeps the separate statistics pass unless VISREP_GN_RESID_256=1 (round 6: built, measured slower, opt-in).  visrep_conv_gn_supported is the round-4 query (no epilogue argument: the
// conservative answer that holds for both).
extern "C" int visrep_conv_gn_supported_epi(int B, int HWo, int Cout, int groups, int epilogue) {
    if (B <= 0 || HWo <= 0 || Cout <= 0 || groups <= 0 || Cout % groups) return 0;
    if (epilogue != VISREP_EPI_BIAS && epilogue != VISREP_EPI_RESID) return 0;
    const int cpg = Cout / groups;
    if (!(HWo % 128 == 0 && Cout % 64 == 0 && (cpg == 4 || cpg == 8 || cpg == 16))) return 0;
    // residual convolutions that the 256x256 kernel takes: round 6 built their partial sums too (gemm_gn_partials_prepass<.., RES>: the pre-pass reads
    // the residual tile as well) and measured the SD1.5 forward 0.4-0.6 ms SLOWER with them than with the separate statistics pass (70.8 / 71.0 ->
    // 71.2 / 71.6 ms, tools/diag/ab_gn_resid256.sh: the second fetch of the residual tile + 7 spilled registers cost what the pass saved) - opt-in:
    // VISREP_GN_RESID_256=1 (read per call, so that tests can switch it)
    const char* e_ = getenv("VISREP_GN_RESID_256");
    const bool resid256 = e_ && e_[0] == '1';
    if (!resid256 && epilogue != VISREP_EPI_BIAS && Cout % 256 == 0 && ((long)B * HWo + 255) / 256 * (Cout / 256) >= 2L * visrep_cu_count()) return 0;
    return 1;
}
extern "C" int visrep_conv_gn_supported(int B, int HWo, int Cout, int groups) {
    return visrep_conv_gn_supported_epi(B, HWo, Cout, groups, VISREP_EPI_RESID);
}

extern "C" size_t visrep_conv_gn_partial_bytes(int B, int HWo, int groups) {
    return (B > 0 && HWo > 0 && groups > 0) ? (size_t)B * (HWo / 64) * groups * sizeof(float2) : 0;
}

extern "C" int visrep_conv3x3_bf16_gn(const void* x, int B, int H, int W, int C, const void* Wt, int ldw, const float* bias, void* out, int ldc,
                                      int Cout, int stride, int pad_mode, int epilogue, const void* resid, void* gn_partial, int groups, void* stream) {
    if (!x || !Wt || !out || !gn_partial) return visrep_set_error(VISREP_ERR_ARG, "conv3x3_gn: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_gn: empty problem");
    if (C % 64) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_gn: C must be a multiple of 64");
    if ((stride != 1 && stride != 2) || (pad_mode != 0 && pad_mode != 1)) return visrep_set_error(VISREP_ERR_ARG, "conv3x3_gn: stride 1|2, pad_mode 0|1");
    if (epilogue != VISREP_EPI_BIAS && epilogue != VISREP_EPI_RESID) return visrep_set_error(VISREP_ERR_ARG, "conv3x3_gn: epilogue must be BIAS or RESID");
    if (epilogue == VISREP_EPI_RESID && !resid) return visrep_set_error(VISREP_ERR_ARG, "conv3x3_gn: EPI_RESID needs resid");
    const int pad_total = pad_mode == 0 ? 2 : 1;
    GemmArgs a{};
    a.A = (const bf16_t*)x; a.W = (const bf16_t*)Wt; a.C = (bf16_t*)out; a.bias = bias; a.resid = (const bf16_t*)resid;
    a.conv = 1; a.cH = H; a.cW = W; a.cC = C; a.cstride = stride; a.cpad = pad_mode == 0 ? 1 : 0; a.cup = 0;
    a.cHo = (H + pad_total - 3) / stride + 1;
    a.cWo = (W + pad_total - 3) / stride + 1;
    if (!visrep_conv_gn_supported_epi(B, a.cHo * a.cWo, Cout, groups, epilogue))
        return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_gn: needs Ho Wo % 128 == 0, 4 | 8 | 16 channels per group and a kernel that emits the sums for this epilogue (see visrep_conv_gn_supported_epi)");
    a.M = B * a.cHo * a.cWo; a.N = Cout; a.K = 9 * C; a.lda = 8; a.ldw = ldw; a.ldc = ldc; a.epi = epilogue;
    a.gn_partial = (float2*)gn_partial; a.gn_cpg = Cout / groups; a.gn_hw = a.cHo * a.cWo;
    return visrep_gemm_dispatch(a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ ViT forward
namespace {
inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct Ws {
    size_t h, qk, vt, vcls, mlp, rt, part, total;
    int ldvt;
};
Ws layout(const visrep_vit_config* c, int B) {
    Ws w;
    const size_t M = (size_t)B * c->tokens;
    const size_t Mp = up(M, 128);
    w.ldvt = (int)(up(M, 64) + 64);
    size_t off = 0;
    w.h = off;   off += up(Mp * c->d * 2, 256);
    w.qk = off;  off += up(Mp * 2 * c->d * 2, 256);
    w.vt = off;  off += up((size_t)c->d * w.ldvt * 2, 256);
    w.vcls = off; off += up((size_t)B * c->d * 2, 256);              // V rows of the CLS tokens (image-aligned attention)
    const size_t mlp_b = Mp * c->mlp * 2;
    const size_t cols_b = up((size_t)B * (c->tokens - c->has_cls), 128) * c->kpad * 2;
    w.mlp = off; off += up(mlp_b > cols_b ? mlp_b : cols_b, 256);   // im2col columns alias the MLP buffer
    w.rt = off;  off += up((Mp + 8) * sizeof(float2), 256);         // folded-LayerNorm row statistics (zero past M)
    w.part = off; off += up(Mp * (c->d / 64) * sizeof(float2), 256);  // per-row partial sums left by the residual GEMM epilogues
    w.total = off;
    return w;
}
}  // namespace

extern "C" size_t visrep_vit_workspace_bytes(const visrep_vit_config* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return layout(cfg, B).total;
}

#define VR_TRY(x) do { const int rc_ = (x); if (rc_) return rc_; } while (0)

extern "C" int visrep_vit_forward(const visrep_vit_config* c, const visrep_vit_weights* w, const void* pixels, int pixel_dtype,
                                  void* hidden, int B, int n_layers, void* workspace, void* stream) {
    if (!c || !w || !pixels || !hidden || !workspace) return visrep_set_error(VISREP_ERR_ARG, "vit_forward: null pointer");
    if (B <= 0) return 0;
    if (n_layers < 0 || n_layers > c->layers) return visrep_set_error(VISREP_ERR_ARG, "vit_forward: n_layers out of range");
    if (c->d != c->heads * 64) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward: head_dim must be 64");
    if (c->d % 128 || c->mlp % 128 || c->kpad % 64) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward: d, mlp % 128 and kpad % 64 required");
    const int grid = c->image_size / c->patch;
    if (grid * grid + (c->has_cls ? 1 : 0) != c->tokens) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward: tokens != grid^2 + cls");
    hipStream_t s = (hipStream_t)stream;
    const Ws L = layout(c, B);
    char* base = (char*)workspace;
    bf16_t* x = (bf16_t*)hidden;
    bf16_t* h = (bf16_t*)(base + L.h);
    bf16_t* qk = (bf16_t*)(base + L.qk);
    bf16_t* vt = (bf16_t*)(base + L.vt);
    bf16_t* vcls = (bf16_t*)(base + L.vcls);
    bf16_t* mlp = (bf16_t*)(base + L.mlp);
    const int d = c->d, T = c->tokens, P = grid * grid, M = B * T;

    // ---- embeddings: conv(k = s = patch) as im2col + GEMM with fused (bias, position add, CLS row skip)
    VR_TRY(visrep_im2col(pixels, pixel_dtype, mlp, B, c->image_size, c->image_size, c->patch, c->kpad, stream));
    GemmArgs g{};
    g.A = mlp; g.lda = c->kpad; g.W = (const bf16_t*)w->patch_w; g.ldw = c->kpad; g.C = x; g.ldc = d;
    g.bias = w->patch_b; g.pos = w->pos; g.M = B * P; g.N = d; g.K = c->kpad; g.epi = EPI_PATCH;
    g.patches = P; g.tokens = T; g.cls_off = c->has_cls ? 1 : 0;
    VR_TRY(visrep_gemm_dispatch(g, s));
    if (c->has_cls) VR_TRY(visrep_cls_rows(x, d, w->cls, w->pos, B, T, d, stream));
    if (c->pre_ln) VR_TRY(visrep_layernorm(x, d, w->pre_ln_g, w->pre_ln_b, x, d, M, d, c->eps, stream));

    // V^T columns past B*T are read (with zero softmax weight) by the last key tile: keep them finite
    if (hipMemsetAsync(vt, 0, (size_t)d * L.ldvt * 2, s) != hipSuccess) return visrep_set_error(VISREP_ERR_LAUNCH, "vit_forward: memset failed");

    float2* rt = (float2*)(base + L.rt);
    if (hipMemsetAsync(rt, 0, (up(M, 128) + 8) * sizeof(float2), s) != hipSuccess) return visrep_set_error(VISREP_ERR_LAUNCH, "vit_forward: memset failed");

    float2* part = (float2*)(base + L.part);
    const float scale = c->q_prescaled ? 0.f : 0.125f;   // head_dim^-0.5, head_dim = 64; 0 = folded into the Q weights (attn_fwd PS)
    // Image-aligned attention (attn_fwd_cls): the patch tokens of an image are whole key tiles and the CLS key is a side term, so V is
    // projected by two row-mapped GEMMs - the patch rows into V^T (column b (T - 1) + t - 1), the CLS rows into vcls [B, d]
    const bool aligned = c->q_prescaled >= 2 && c->has_cls && visrep_mhsa_cls_supported(T) && t_visrep_attn_variant == 1;
    bool rt_ready = false;        // rt already holds the statistics of x (left by the previous layer's fc2 GEMM)
    for (int l = 0; l < n_layers; ++l) {
        const visrep_vit_layer& W = w->layers[l];
        const bool fold = W.sqkv && W.s1;     // LayerNorm folded into the QK / V / fc1 GEMMs: only its statistics are computed,
        const bool fold_next = l + 1 < n_layers && w->layers[l + 1].sqkv && w->layers[l + 1].s1;   // and those by the GEMM that writes x
        GemmArgs a{};
        if (fold) {
            if (!rt_ready) VR_TRY(visrep_layernorm_stats(x, d, rt, M, d, c->eps, stream));
            a.A = x; a.ln_rt = rt; a.ln_s = W.sqkv;
        } else {
            VR_TRY(visrep_layernorm(x, d, W.ln1_g, W.ln1_b, h, d, M, d, c->eps, stream));
            a.A = h;
        }
        a.lda = d; a.K = d; a.M = M;
        // Q | K projection
        a.W = (const bf16_t*)W.wqkv; a.ldw = d; a.N = 2 * d; a.C = qk; a.ldc = 2 * d; a.bias = W.bqkv; a.epi = EPI_BIAS;
        VR_TRY(visrep_gemm_dispatch(a, s));
        // V projection written transposed + perm16 for the attention kernel
        a.W = (const bf16_t*)W.wqkv + (size_t)2 * d * d; a.N = d; a.C = vt; a.ldc = L.ldvt; a.bias = W.bqkv + 2 * d; a.epi = EPI_VT;
        if (fold) a.ln_s = W.sqkv + 2 * d;
        if (aligned) {
            GemmArgs v = a;
            v.M = B * (T - 1); v.a_period = T - 1; v.a_stride = T; v.a_first = 1;
            VR_TRY(visrep_gemm_dispatch(v, s));
            v.M = B; v.a_period = 1; v.a_first = 0; v.C = vcls; v.ldc = d; v.epi = EPI_BIAS;
            VR_TRY(visrep_gemm_dispatch(v, s));
            VR_TRY(visrep_mhsa_cls_fwd(qk, 2 * d, vt, L.ldvt, vcls, d, h, d, B, T, c->heads, 64, stream));
        } else {
            VR_TRY(visrep_gemm_dispatch(a, s));
            VR_TRY(visrep_mhsa_fwd(qk, 2 * d, vt, L.ldvt, h, d, B, T, c->heads, 64, scale, stream));
        }
        // out projection + LayerScale + residual (in place on x)
        a.A = h; a.ln_rt = nullptr; a.ln_s = nullptr;
        a.W = (const bf16_t*)W.wo; a.N = d; a.C = x; a.ldc = d; a.bias = W.bo; a.epi = EPI_RESID; a.resid = x; a.ls = W.ls1;
        if (fold) { a.stat_rt = rt; a.stat_partial = part; a.stat_eps = c->eps; }   // LN2's statistics come out of this GEMM
        VR_TRY(visrep_gemm_dispatch(a, s));
        GemmArgs f{};
        if (fold) {
            f.A = x; f.ln_rt = rt; f.ln_s = W.s1;
        } else {
            VR_TRY(visrep_layernorm(x, d, W.ln2_g, W.ln2_b, h, d, M, d, c->eps, stream));
            f.A = h;
        }
        f.lda = d; f.K = d; f.M = M; f.W = (const bf16_t*)W.w1; f.ldw = d; f.N = c->mlp; f.C = mlp; f.ldc = c->mlp;
        f.bias = W.b1; f.epi = EPI_ACT; f.act = c->act;
        VR_TRY(visrep_gemm_dispatch(f, s));
        f.ln_rt = nullptr; f.ln_s = nullptr;
        f.A = mlp; f.lda = c->mlp; f.K = c->mlp; f.W = (const bf16_t*)W.w2; f.ldw = c->mlp; f.N = d; f.C = x; f.ldc = d;
        f.bias = W.b2; f.epi = EPI_RESID; f.act = 0; f.resid = x; f.ls = W.ls2;
        if (fold_next) { f.stat_rt = rt; f.stat_partial = part; f.stat_eps = c->eps; }   // the next layer's LN1
        rt_ready = fold_next;
        VR_TRY(visrep_gemm_dispatch(f, s));
    }
    return 0;
}
