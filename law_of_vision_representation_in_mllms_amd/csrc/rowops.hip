// Row-wise / gather kernels around the GEMMs: LayerNorm, patch im2col, CLS+pos row, casts.  All HBM-bound:
// 16-byte vector accesses, one wave per row for the reductions (wave64 shuffles, no LDS).
#include "common.h"
#include "visrep_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm
// y[row] = (x[row] - mean) * rsqrt(var + eps) * gamma + beta   (two-pass over registers, fp32 statistics).
// One wave per row, d <= 2048, d % 8 == 0.  In-place (y == x) is safe: a lane reads all its chunks before writing.
__global__ __launch_bounds__(256) void layernorm_rows(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ g,
                                                      const float* __restrict__ bta, bf16_t* __restrict__ y, int ldy,
                                                      int M, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = d >> 3;
    const bf16_t* xr = x + (size_t)row * ldx;
    float f[4][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + ch * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { f[c][2 * k] = bf_lo(v[k]); f[c][2 * k + 1] = bf_hi(v[k]); }
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += f[c][k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[c][k] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (lane + 64 * c < nch) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float t = f[c][k] - mean; sq += t * t; }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)d + eps);
    bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const float4 g0 = *reinterpret_cast<const float4*>(g + ch * 8), g1 = *reinterpret_cast<const float4*>(g + ch * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(bta + ch * 8), b1 = *reinterpret_cast<const float4*>(bta + ch * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (f[c][k] - mean) * rstd * gg[k] + bb[k];
            u32x4 w = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
            *reinterpret_cast<u32x4*>(yr + ch * 8) = w;
        }
    }
}

// LayerNorm statistics only: rt[row] = (rstd, -mean * rstd).  The normalisation itself is folded into the consuming GEMM
// (gamma into its weights, the mean / rstd terms into its epilogue: visrep_gemm_bf16_ln), so the normalised tensor is never
// written: this pass only READS the residual stream.  One wave per row, same two-pass fp32 statistics as layernorm_rows.
__global__ __launch_bounds__(256) void layernorm_stats_rows(const bf16_t* __restrict__ x, int ldx, float2* __restrict__ rt, int M, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = d >> 3;
    const bf16_t* xr = x + (size_t)row * ldx;
    float f[4][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + ch * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { f[c][2 * k] = bf_lo(v[k]); f[c][2 * k + 1] = bf_hi(v[k]); }
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += f[c][k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[c][k] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (lane + 64 * c < nch) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float t = f[c][k] - mean; sq += t * t; }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)d + eps);
    if (lane == 0) rt[row] = float2{rstd, -mean * rstd};
}

// Row statistics from the partial sums a residual GEMM's epilogue left behind (gemm_epilogue.h, HAS_ST): slot k of row m holds
// (sum, sum of squares) of 64 output columns; summed here in slot order (deterministic), var = E[x^2] - mean^2 in fp32.
__global__ __launch_bounds__(256) void ln_stats_finalize_rows(const float2* __restrict__ part, int slots, float2* __restrict__ rt, int M,
                                                              float inv_d, float eps) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    const float4* src = reinterpret_cast<const float4*>(part + (size_t)row * slots);   // slots is even: two slots per 16-B load
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < slots / 2; ++k) {
        const float4 v = src[k];
        s1 += v.x; s2 += v.y;
        s1 += v.z; s2 += v.w;
    }
    const float mean = s1 * inv_d;
    const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    rt[row] = float2{rstd, -mean * rstd};
}

// ------------------------------------------------------------------------------------------------ im2col
// pixels [B, 3, H, W] (fp32 or bf16) -> cols [B*gh*gw, Kpad] bf16 with k = c*p*p + i*p + j (Conv2d weight order),
// zero-filled for k >= 3*p*p.  One thread per 8 output elements (16-B store).
template <typename TIN>
__global__ __launch_bounds__(256) void im2col_patches(const TIN* __restrict__ px, bf16_t* __restrict__ cols, int B, int Himg,
                                                      int Wimg, int patch, int Kpad) {
    const int gh = Himg / patch, gw = Wimg / patch;
    const int kch = Kpad >> 3;
    const long total = (long)B * gh * gw * kch;
    const int K = 3 * patch * patch, pp = patch * patch;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % kch);
        const long row = idx / kch;
        const int gx = (int)(row % gw);
        const int gy = (int)((row / gw) % gh);
        const int b = (int)(row / ((long)gw * gh));
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            float val = 0.f;
            if (k < K) {
                const int c = k / pp, rem = k - c * pp;
                const int i = rem / patch, j = rem - i * patch;
                const size_t off = (((size_t)b * 3 + c) * Himg + (gy * patch + i)) * Wimg + gx * patch + j;
                if (sizeof(TIN) == 4) val = reinterpret_cast<const float*>(px)[off];
                else val = bf2f(reinterpret_cast<const bf16_t*>(px)[off]);
            }
            v[e] = val;
        }
        u32x4 w = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
        *reinterpret_cast<u32x4*>(cols + row * Kpad + ch * 8) = w;
    }
}

// x[b*T + 0][:] = cls + pos[0]
__global__ void cls_rows(bf16_t* __restrict__ x, int ldx, const float* __restrict__ cls, const float* __restrict__ pos,
                         int B, int T, int d) {
    const int b = blockIdx.x;
    for (int n = threadIdx.x; n < d; n += blockDim.x) x[(size_t)b * T * ldx + n] = f2bf(cls[n] + pos[n]);
}

__global__ void cast_f32_to_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (long)gridDim.x * blockDim.x * 2) {
        if (i + 1 < n) *reinterpret_cast<uint32_t*>(dst + i) = pack_bf16(src[i], src[i + 1]);
        else dst[i] = f2bf(src[i]);
    }
}

}  // namespace

extern "C" int visrep_layernorm(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int rows,
                                int d, float eps, void* stream) {
    if (rows <= 0) return 0;
    if (d % 8 || d > 2048 || (ldx % 8) || (ldy % 8)) return visrep_set_error(VISREP_ERR_SHAPE, "layernorm: need d % 8 == 0, d <= 2048");
    hipLaunchKernelGGL(layernorm_rows, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, gamma, beta,
                       (bf16_t*)y, ldy, rows, d, eps);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "layernorm: launch failed");
}

extern "C" int visrep_layernorm_stats(const void* x, int ldx, void* rt, int rows, int d, float eps, void* stream) {
    if (rows <= 0) return 0;
    if (d % 8 || d > 2048 || (ldx % 8)) return visrep_set_error(VISREP_ERR_SHAPE, "layernorm_stats: need d % 8 == 0, d <= 2048");
    hipLaunchKernelGGL(layernorm_stats_rows, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (float2*)rt, rows, d, eps);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "layernorm_stats: launch failed");
}

int visrep_ln_stats_finalize(const float2* partial, int slots, float2* rt, int rows, int d, float eps, hipStream_t s) {
    if (rows <= 0) return 0;
    if (slots <= 0 || (slots & 1)) return visrep_set_error(VISREP_ERR_SHAPE, "ln_stats_finalize: slot count must be even");
    hipLaunchKernelGGL(ln_stats_finalize_rows, dim3((rows + 255) / 256), dim3(256), 0, s, partial, slots, rt, rows, 1.0f / (float)d, eps);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "ln_stats_finalize: launch failed");
}

extern "C" int visrep_im2col(const void* pixels, int pixel_dtype, void* cols, int B, int Himg, int Wimg, int patch, int Kpad,
                             void* stream) {
    if (B <= 0) return 0;
    if (Himg % patch || Wimg % patch || Kpad % 8 || Kpad < 3 * patch * patch)
        return visrep_set_error(VISREP_ERR_SHAPE, "im2col: image must be a multiple of the patch, Kpad >= 3*p*p and % 8");
    const long total = (long)B * (Himg / patch) * (Wimg / patch) * (Kpad / 8);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (pixel_dtype == VISREP_F32)
        hipLaunchKernelGGL(im2col_patches<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)pixels, (bf16_t*)cols, B,
                           Himg, Wimg, patch, Kpad);
    else if (pixel_dtype == VISREP_BF16)
        hipLaunchKernelGGL(im2col_patches<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pixels, (bf16_t*)cols, B,
                           Himg, Wimg, patch, Kpad);
    else
        return visrep_set_error(VISREP_ERR_ARG, "im2col: pixel dtype must be VISREP_F32 or VISREP_BF16");
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "im2col: launch failed");
}

extern "C" int visrep_cls_rows(void* x, int ldx, const float* cls, const float* pos, int B, int T, int d, void* stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(cls_rows, dim3(B), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ldx, cls, pos, B, T, d);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "cls_rows: launch failed");
}

extern "C" int visrep_cast_f32_bf16(const float* src, void* dst, long n, void* stream) {
    if (n <= 0) return 0;
    long blocks = (n / 2 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_f32_to_bf16, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "cast: launch failed");
}
