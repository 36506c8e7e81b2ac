// Convolutional-tower primitives of the diffusion feature extractors (SD UNet + VAE encoder) on gfx950.
//
// Data layout: every activation is CHANNELS-LAST, token-major bf16 - a [B*H*W, C] matrix, the same thing the GEMM and
// attention kernels consume.  1x1 convolutions and Linear layers are then plain GEMMs, the transformer blocks need no
// NCHW<->token permutes (the reference does two per block, transformer_2d.py:145,170), and a 3x3 convolution is
// gather (this file) + GEMM with K = 9*C ordered (tap, channel), weights repacked [Cout, 3, 3, Cin] once at load time.
//
// Everything here is HBM-bound element/gather work: 16-byte accesses, coalesced along the channel axis.
//   groupnorm_stats / groupnorm_apply   GroupNorm(+SiLU) over (H*W, C/groups) per (image, group)     resnet.py:ResnetBlock2D
//   im2col3x3                           3x3 patch gather: stride 1|2, symmetric or (0,1,0,1) padding, optional fused
//                                       nearest-2x upsample of the source (upsampling.py Upsample2D, downsampling.py)
//   geglu                               val * gelu(gate)                                              activations.py GEGLU
//   softmax_rows                        fp32 scores -> bf16 probabilities (single-head VAE attention)
//   nchw_to_tokens                      [B,C,H,W] fp32|bf16 -> [B*H*W, Cpad] bf16, zero-padded channels
//   sd_noisy_latents                    posterior sample * scaling_factor, then DDIM add_noise       dift_sd.py:172-176
//   mean_groups                         ensemble mean                                                 dift_sd.py:275
#include "common.h"
#include "visrep_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------ GroupNorm
// Deterministic two-level reduction (no atomics: the towers are deep enough that one flipped bf16 rounding in a statistic
// moves the final features by 1e-2, so run-to-run reproducibility needs a fixed summation order):
//   groupnorm_stats     block (row chunk, image): threads [R row lanes][W channel-pair lanes] -> per-thread partial sums in
//                       LDS -> thread g adds its group's entries in a fixed order -> partial[(b*nblk + blk)*G + g] = (s, ss)
//   groupnorm_finalize  thread (b, g): adds the nblk partials in order -> stats[b*G + g] = (mean, rstd)
//   groupnorm_apply     one 16-byte vector per thread
__global__ __launch_bounds__(256) void groupnorm_stats(const bf16_t* __restrict__ x, float2* __restrict__ partial, int HW, int C, int cpg,
                                                       int rows_per_block, int W) {
    extern __shared__ float2 buf[];                            // [R][npair]
    const int tid = threadIdx.x, b = blockIdx.y, G = C / cpg;
    const int tc = tid % W, tr = tid / W, R = 256 / W;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    const uint32_t* base = reinterpret_cast<const uint32_t*>(x + (size_t)b * HW * C);
    const int npair = C >> 1;
    for (int c2 = tc; c2 < npair; c2 += W) {
        float s = 0.f, ss = 0.f;
        int r = r0 + tr;
        for (; r + 3 * R < r1; r += 4 * R) {                   // four independent loads in flight
            const uint32_t v0 = base[(size_t)r * npair + c2], v1 = base[(size_t)(r + R) * npair + c2];
            const uint32_t v2 = base[(size_t)(r + 2 * R) * npair + c2], v3 = base[(size_t)(r + 3 * R) * npair + c2];
            const float a0 = bf_lo(v0), a1 = bf_hi(v0), a2 = bf_lo(v1), a3 = bf_hi(v1);
            const float a4 = bf_lo(v2), a5 = bf_hi(v2), a6 = bf_lo(v3), a7 = bf_hi(v3);
            s += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
            ss += ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((a4 * a4 + a5 * a5) + (a6 * a6 + a7 * a7));
        }
        for (; r < r1; r += R) {
            const uint32_t v = base[(size_t)r * npair + c2];
            const float a0 = bf_lo(v), a1 = bf_hi(v);
            s += a0 + a1;
            ss += a0 * a0 + a1 * a1;
        }
        buf[tr * npair + c2] = float2{s, ss};
    }
    __syncthreads();
    const int ppg = cpg >> 1;                                   // channel pairs per group (cpg is even)
    for (int g = tid; g < G; g += 256) {
        float s = 0.f, ss = 0.f;
        for (int rr = 0; rr < R; ++rr)
            for (int k = 0; k < ppg; ++k) {
                const float2 v = buf[rr * npair + g * ppg + k];
                s += v.x;
                ss += v.y;
            }
        partial[((size_t)b * gridDim.x + blockIdx.x) * G + g] = float2{s, ss};
    }
}

__global__ __launch_bounds__(256) void groupnorm_finalize(const float2* __restrict__ partial, float2* __restrict__ stats, int BG, int G, int nblk,
                                                          float inv_n, float eps) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);         // one wave per (b, g): i = b*G + g
    if (i >= BG) return;
    const int b = i / G, g = i - b * G;
    float s = 0.f, ss = 0.f;
    for (int k = lane; k < nblk; k += 64) {                    // lane-strided, then a fixed butterfly: order never changes
        const float2 v = partial[((size_t)b * nblk + k) * G + g];
        s += v.x;
        ss += v.y;
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) {
        const float mean = s * inv_n;
        const float var = fmaxf(ss * inv_n - mean * mean, 0.f);
        stats[i] = float2{mean, rsqrtf(var + eps)};
    }
}

__global__ __launch_bounds__(256) void groupnorm_apply(const bf16_t* __restrict__ x, const float2* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16_t* __restrict__ y, long total_vec, int HW, int C, int cpg, int silu) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total_vec) return;
    const int cv8 = C >> 3;
    const long row = idx / cv8;
    const int c0 = (int)(idx - row * cv8) * 8;
    const int b = (int)(row / HW), G = C / cpg;
    const u32x4 raw = *reinterpret_cast<const u32x4*>(x + row * C + c0);
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(raw[e]); v[2 * e + 1] = bf_hi(raw[e]); }
    int gprev = -1;
    float2 st = float2{0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c0 + e) / cpg;
        if (g != gprev) { st = stats[(size_t)b * G + g]; gprev = g; }
        float o = (v[e] - st.x) * st.y * gm[e] + bt[e];
        if (silu) o = o * __builtin_amdgcn_rcpf(1.0f + __expf(-o));
        v[e] = o;
    }
    u32x4 out = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
    *reinterpret_cast<u32x4*>(y + row * C + c0) = out;
}

// Row-streaming form for C / 8 a power of two <= 256 (the VAE's 128 / 256 / 512 channels, the transformer widths): a thread keeps ONE 8-channel
// vector position for the whole launch - its (scale, shift) pairs (rstd gamma, beta - mean rstd gamma) are built once from 8 gamma, 8 beta
// and the one or two group statistics - and walks the rows of its image chunk with four 16-byte loads in flight.  groupnorm_apply above
// re-derives row, image, group (integer divisions by run-time values) and re-loads 80 bytes of parameters for every 16 bytes of payload:
// 3.85 TB/s of read + write traffic at 768^2 x 128 channels against ~5 here (round 4, profiles/round4_sd15_kernel_stats.md).
template <bool SILU>
__global__ __launch_bounds__(256) void groupnorm_apply_rows(const bf16_t* __restrict__ x, const float2* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16_t* __restrict__ y, int HW, int C, int cpg, int rows_per_block) {
    const int cv8 = C >> 3, R = 256 / cv8;
    const int tc = threadIdx.x & (cv8 - 1), tr = threadIdx.x / cv8;
    const int b = blockIdx.y, G = C / cpg, c0 = tc * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float2 st = stats[(size_t)b * G + (c0 + e) / cpg];
        const float gsc = st.y * gamma[c0 + e];
        sc[e] = gsc;
        sh[e] = __builtin_fmaf(-st.x, gsc, beta[c0 + e]);
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    const size_t base = (size_t)b * HW * C + c0;
    auto norm = [&](const u32x4 raw) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(raw[e]); v[2 * e + 1] = bf_hi(raw[e]); }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = __builtin_fmaf(v[e], sc[e], sh[e]);
            if (SILU) o = o * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(o * -1.4426950408889634f));
            v[e] = o;
        }
        return u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
    };
    int r = r0 + tr;
    for (; r + 3 * R < r1; r += 4 * R) {
        u32x4 raw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) raw[k] = *reinterpret_cast<const u32x4*>(x + base + (size_t)(r + k * R) * C);
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(y + base + (size_t)(r + k * R) * C) = norm(raw[k]);
    }
    for (; r < r1; r += R) *reinterpret_cast<u32x4*>(y + base + (size_t)r * C) = norm(*reinterpret_cast<const u32x4*>(x + base + (size_t)r * C));
}

// ------------------------------------------------------------------------------------------------ 3x3 gather
struct Im2colArgs {
    const bf16_t* x; bf16_t* y;
    int B, H, W, C, Ho, Wo, stride, pad_lo, up, ldy;            // H, W: source size BEFORE the optional 2x upsample
};

__global__ __launch_bounds__(256) void im2col3x3(const Im2colArgs p, long total_vec) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total_vec) return;
    const int vec_per_row = p.ldy >> 3, cv8 = p.C >> 3;
    const long orow = idx / vec_per_row;
    const int v = (int)(idx - orow * vec_per_row);
    u32x4 val = {0u, 0u, 0u, 0u};
    if (v < 9 * cv8) {
        const int tap = v / cv8, c0 = (v - tap * cv8) * 8;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int hw = p.Ho * p.Wo;
        const int b = (int)(orow / hw), pix = (int)(orow - (long)b * hw);
        const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
        int iy = oy * p.stride + ky - p.pad_lo, ix = ox * p.stride + kx - p.pad_lo;
        const int Hl = p.H << p.up, Wl = p.W << p.up;          // logical (upsampled) source size
        if (iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) {
            iy >>= p.up; ix >>= p.up;                           // nearest neighbour: floor(i / 2)
            val = *reinterpret_cast<const u32x4*>(p.x + (((size_t)b * p.H + iy) * p.W + ix) * p.C + c0);
        }
    }
    *reinterpret_cast<u32x4*>(p.y + orow * p.ldy + (size_t)v * 8) = val;
}

// ------------------------------------------------------------------------------------------------ GEGLU
__global__ __launch_bounds__(256) void geglu(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long total_vec, int F, int ldx, int ldy) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total_vec) return;
    const int fv = F >> 3;
    const long row = idx / fv;
    const int f0 = (int)(idx - row * fv) * 8;
    const u32x4 a = *reinterpret_cast<const u32x4*>(x + row * ldx + f0);
    const u32x4 g = *reinterpret_cast<const u32x4*>(x + row * ldx + F + f0);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g0 = bf_lo(g[e]), g1 = bf_hi(g[e]);
        const float h0 = gelu_erf(g0), h1 = gelu_erf(g1);
        o[e] = pack_bf16(bf_lo(a[e]) * h0, bf_hi(a[e]) * h1);
    }
    *reinterpret_cast<u32x4*>(y + row * ldy + f0) = o;
}

// ------------------------------------------------------------------------------------------------ row softmax (fp32 -> bf16)
__global__ __launch_bounds__(256) void softmax_rows(const float* __restrict__ s, bf16_t* __restrict__ pr, int n, int lds_, int ldp, float scale) {
    __shared__ float red[4];
    const float* row = s + (size_t)blockIdx.x * lds_;
    bf16_t* out = pr + (size_t)blockIdx.x * ldp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, row[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float sc = scale * 1.4426950408889634f;
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) sum += __builtin_amdgcn_exp2f((row[i] - m) * sc);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = tid; i < ldp; i += 256) {                       // pad columns [n, ldp) are written as zeros (GEMM K padding)
        const float v = i < n ? __builtin_amdgcn_exp2f((row[i] - m) * sc) * inv : 0.f;
        out[i] = (bf16_t)(pack_bf16(v, 0.f) & 0xffffu);
    }
}

// ------------------------------------------------------------------------------------------------ layout / latent helpers
template <bool F32>
VR_DEV float ld_elem(const void* x, long i) {
    if (F32) return reinterpret_cast<const float*>(x)[i];
    return __uint_as_float((uint32_t) reinterpret_cast<const bf16_t*>(x)[i] << 16);
}

template <bool F32>
__global__ __launch_bounds__(256) void nchw_to_tokens(const void* __restrict__ x, bf16_t* __restrict__ y, long total, int C, int HW, int Cpad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // one thread per (b, pixel)
    if (idx >= total) return;
    const long b = idx / HW;
    const int pix = (int)(idx - b * HW);
    bf16_t* out = y + idx * Cpad;
    for (int c = 0; c < Cpad; c += 2) {
        const float v0 = c < C ? ld_elem<F32>(x, (b * C + c) * HW + pix) : 0.f;
        const float v1 = c + 1 < C ? ld_elem<F32>(x, (b * C + c + 1) * HW + pix) : 0.f;
        *reinterpret_cast<uint32_t*>(out + c) = pack_bf16(v0, v1);
    }
}

__global__ __launch_bounds__(256) void sd_noisy_latents(const float* __restrict__ moments, int ldm, const float* __restrict__ post,
                                                        const float* __restrict__ ddim, bf16_t* __restrict__ y, long total, int Z, int HW,
                                                        int Cpad, float scaling, float sqrt_ac, float sqrt_1mac) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // one thread per (b, pixel)
    if (idx >= total) return;
    const long b = idx / HW;
    const int pix = (int)(idx - b * HW);
    const float* mrow = moments + idx * ldm;
    bf16_t* out = y + idx * Cpad;
    for (int c = 0; c < Cpad; c += 2) {
        float v[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ch = c + e;
            if (ch < Z) {
                const float mean = mrow[ch], logvar = fminf(fmaxf(mrow[Z + ch], -30.0f), 20.0f);
                const long ni = (b * Z + ch) * HW + pix;
                const float lat = (mean + __expf(0.5f * logvar) * post[ni]) * scaling;
                v[e] = sqrt_ac * lat + sqrt_1mac * ddim[ni];
            }
        }
        *reinterpret_cast<uint32_t*>(out + c) = pack_bf16(v[0], v[1]);
    }
}

__global__ __launch_bounds__(256) void mean_groups(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long total_pairs, long n_pairs, int E) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // one thread per output bf16 pair
    if (idx >= total_pairs) return;
    const long b = idx / n_pairs, i = idx - b * n_pairs;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(x) + b * E * n_pairs + i;
    float s0 = 0.f, s1 = 0.f;
    for (int e = 0; e < E; ++e) { const uint32_t v = src[(long)e * n_pairs]; s0 += bf_lo(v); s1 += bf_hi(v); }
    const float inv = 1.0f / (float)E;
    reinterpret_cast<uint32_t*>(y)[idx] = pack_bf16(s0 * inv, s1 * inv);
}

// F.interpolate(mode="bilinear", align_corners=False, antialias=False) on NCHW planes, fp32|bf16 in -> bf16 out
template <bool F32>
__global__ __launch_bounds__(256) void resize_bilinear(const void* __restrict__ x, bf16_t* __restrict__ y, long total, int H, int W, int OH, int OW,
                                                       float sh, float sw) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // one thread per output element
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const long r = idx / OW;
    const int oy = (int)(r % OH);
    const long plane = r / OH;                                 // b * C + c
    const float sy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - y0, lx = sx - x0;
    const long base = plane * H * W;
    const float v00 = ld_elem<F32>(x, base + (long)y0 * W + x0), v01 = ld_elem<F32>(x, base + (long)y0 * W + x1);
    const float v10 = ld_elem<F32>(x, base + (long)y1 * W + x0), v11 = ld_elem<F32>(x, base + (long)y1 * W + x1);
    const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    y[idx] = (bf16_t)(pack_bf16(v, 0.f) & 0xffffu);
}

inline unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }
inline int launched(const char* what) {
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, what);
}

}  // namespace

namespace {
inline int gn_rows_per_block(int B, int HW) {
    // enough blocks to fill 256 CUs a few times over, but at least 32 rows per block
    int rows = (int)(((long)B * HW + 2047) / 2048);
    rows = rows < 32 ? 32 : rows;
    return rows > HW ? HW : rows;
}
}  // namespace

namespace {
int launch_groupnorm_apply(const void* x, const float2* stats, const float* gamma, const float* beta, void* y, int B, int HW, int C, int cpg, int silu,
                           hipStream_t st) {
    const int cv8 = C / 8;
    if (cv8 <= 256 && (cv8 & (cv8 - 1)) == 0) {                 // row-streaming form; 64 rows per thread unless the image is small
        const int Ra = 256 / cv8;
        int arows = 64 * Ra;
        while (arows > 4 * Ra && (long)((HW + arows - 1) / arows) * B < 2048) arows >>= 1;      // keep >= 8 blocks per CU in flight
        const dim3 grid((HW + arows - 1) / arows, B);
        if (silu) hipLaunchKernelGGL(groupnorm_apply_rows<true>, grid, dim3(256), 0, st, (const bf16_t*)x, (const float2*)stats, gamma, beta, (bf16_t*)y, HW, C, cpg, arows);
        else hipLaunchKernelGGL(groupnorm_apply_rows<false>, grid, dim3(256), 0, st, (const bf16_t*)x, (const float2*)stats, gamma, beta, (bf16_t*)y, HW, C, cpg, arows);
        return launched("groupnorm: launch failed");
    }
    const long total = (long)B * HW * (C / 8);
    hipLaunchKernelGGL(groupnorm_apply, dim3(blocks_for(total)), dim3(256), 0, st, (const bf16_t*)x, (const float2*)stats, gamma, beta,
                       (bf16_t*)y, total, HW, C, cpg, silu);
    return launched("groupnorm: launch failed");
}
}  // namespace

extern "C" size_t visrep_groupnorm_workspace_bytes(int B, int HW, int groups) {
    if (B <= 0 || HW <= 0 || groups <= 0) return 0;
    const int rows = gn_rows_per_block(B, HW), nblk = (HW + rows - 1) / rows;
    return (size_t)B * groups * (nblk + 1) * sizeof(float2);
}

extern "C" int visrep_groupnorm(const void* x, const float* gamma, const float* beta, void* y, int B, int HW, int C, int groups,
                                float eps, int silu, void* workspace, void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm: empty problem");
    if (C % groups || C % 8) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm: C must be a multiple of groups and of 8");
    const int cpg = C / groups;
    if (cpg & 1) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm: channels per group must be even");
    if (!workspace) return visrep_set_error(VISREP_ERR_ARG, "groupnorm: workspace missing");
    hipStream_t st = (hipStream_t)stream;
    const int npair = C / 2;
    int W = 256;
    if (npair <= 128) { W = 1; while (W < npair) W <<= 1; }
    const int R = 256 / W, rows = gn_rows_per_block(B, HW), nblk = (HW + rows - 1) / rows;
    float2* stats = (float2*)workspace;
    float2* partial = stats + (size_t)B * groups;
    const size_t lds = (size_t)R * npair * sizeof(float2);
    hipLaunchKernelGGL(groupnorm_stats, dim3(nblk, B), dim3(256), lds, st, (const bf16_t*)x, partial, HW, C, cpg, rows, W);
    const int BG = B * groups;
    hipLaunchKernelGGL(groupnorm_finalize, dim3((BG + 3) / 4), dim3(256), 0, st, (const float2*)partial, stats, BG, groups, nblk,
                       1.0f / ((float)HW * (float)cpg), eps);
    return launch_groupnorm_apply(x, stats, gamma, beta, y, B, HW, C, cpg, silu, st);
}

// The statistics half alone (the fused convolution conv_halo.hip applies the normalisation itself): stats[b, g] = (mean, rstd)
extern "C" int visrep_groupnorm_stats(const void* x, void* stats_out, int B, int HW, int C, int groups, float eps, void* workspace, void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm_stats: empty problem");
    if (C % groups || C % 8 || ((C / groups) & 1)) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm_stats: C % groups, C % 8, even channels per group");
    if (!x || !stats_out || !workspace) return visrep_set_error(VISREP_ERR_ARG, "groupnorm_stats: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int cpg = C / groups, npair = C / 2;
    int W = 256;
    if (npair <= 128) { W = 1; while (W < npair) W <<= 1; }
    const int R = 256 / W, rows = gn_rows_per_block(B, HW), nblk = (HW + rows - 1) / rows;
    float2* partial = (float2*)workspace + (size_t)B * groups;          // same workspace layout as visrep_groupnorm (its stats slot stays unused)
    hipLaunchKernelGGL(groupnorm_stats, dim3(nblk, B), dim3(256), (size_t)R * npair * sizeof(float2), st, (const bf16_t*)x, partial, HW, C, cpg, rows, W);
    const int BG = B * groups;
    hipLaunchKernelGGL(groupnorm_finalize, dim3((BG + 3) / 4), dim3(256), 0, st, (const float2*)partial, (float2*)stats_out, BG, groups, nblk,
                       1.0f / ((float)HW * (float)cpg), eps);
    return launched("groupnorm_stats: launch failed");
}

extern "C" int visrep_groupnorm_stats_from_partials(const void* partial, void* stats_out, int B, int HW, int C, int groups, float eps, void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups || HW % 64) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm_stats_from_partials: bad shape");
    if (!partial || !stats_out) return visrep_set_error(VISREP_ERR_ARG, "groupnorm_stats_from_partials: null pointer");
    const int cpg = C / groups, BG = B * groups;
    hipLaunchKernelGGL(groupnorm_finalize, dim3((BG + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float2*)partial, (float2*)stats_out, BG, groups, HW / 64,
                       1.0f / ((float)HW * (float)cpg), eps);
    return launched("groupnorm_stats_from_partials: launch failed");
}

// GroupNorm whose per-(image, 64-row slot, group) partial sums came out of the producing convolution's epilogue (visrep_conv3x3_bf16_gn):
// finalize + apply only - the tensor is not read for its statistics.  workspace: >= B * groups float2.
extern "C" int visrep_groupnorm_from_partials(const void* x, const float* gamma, const float* beta, void* y, int B, int HW, int C, int groups, float eps,
                                              int silu, const void* partial, void* workspace, void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm: empty problem");
    if (C % groups || C % 8 || HW % 64) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm_from_partials: C % groups, C % 8, HW % 64 must be 0");
    if (!x || !y || !partial || !workspace) return visrep_set_error(VISREP_ERR_ARG, "groupnorm_from_partials: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int cpg = C / groups, BG = B * groups, nblk = HW / 64;
    float2* stats = (float2*)workspace;
    hipLaunchKernelGGL(groupnorm_finalize, dim3((BG + 3) / 4), dim3(256), 0, st, (const float2*)partial, stats, BG, groups, nblk,
                       1.0f / ((float)HW * (float)cpg), eps);
    return launch_groupnorm_apply(x, stats, gamma, beta, y, B, HW, C, cpg, silu, st);
}

extern "C" int visrep_im2col3x3(const void* x, void* y, int B, int H, int W, int C, int stride, int pad_mode, int upsample, int ldy,
                                void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "im2col3x3: empty problem");
    if (C % 8 || ldy % 8 || ldy < 9 * C) return visrep_set_error(VISREP_ERR_SHAPE, "im2col3x3: C and ldy must be multiples of 8, ldy >= 9*C");
    if ((stride != 1 && stride != 2) || (pad_mode != 0 && pad_mode != 1) || (upsample != 0 && upsample != 1))
        return visrep_set_error(VISREP_ERR_ARG, "im2col3x3: stride 1|2, pad_mode 0 (symmetric 1) | 1 (0,1,0,1), upsample 0|1");
    Im2colArgs a;
    a.x = (const bf16_t*)x; a.y = (bf16_t*)y; a.B = B; a.H = H; a.W = W; a.C = C; a.stride = stride; a.up = upsample; a.ldy = ldy;
    a.pad_lo = pad_mode == 0 ? 1 : 0;
    const int Hl = H << upsample, Wl = W << upsample, pad_total = pad_mode == 0 ? 2 : 1;
    a.Ho = (Hl + pad_total - 3) / stride + 1;
    a.Wo = (Wl + pad_total - 3) / stride + 1;
    const long total = (long)B * a.Ho * a.Wo * (ldy / 8);
    hipLaunchKernelGGL(im2col3x3, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, a, total);
    return launched("im2col3x3: launch failed");
}

extern "C" int visrep_geglu(const void* x, int ldx, void* y, int ldy, long M, int F, void* stream) {
    if (M <= 0 || F <= 0 || F % 8 || ldx % 8 || ldy % 8) return visrep_set_error(VISREP_ERR_SHAPE, "geglu: F and leading dimensions must be multiples of 8");
    const long total = M * (F / 8);
    hipLaunchKernelGGL(geglu, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, total, F, ldx, ldy);
    return launched("geglu: launch failed");
}

extern "C" int visrep_softmax_rows(const float* scores, int lds_, void* probs, int ldp, int rows, int n, float scale, void* stream) {
    if (rows <= 0 || n <= 0 || ldp < n || lds_ < n) return visrep_set_error(VISREP_ERR_SHAPE, "softmax_rows: bad shape");
    hipLaunchKernelGGL(softmax_rows, dim3(rows), dim3(256), 0, (hipStream_t)stream, scores, (bf16_t*)probs, n, lds_, ldp, scale);
    return launched("softmax_rows: launch failed");
}

extern "C" int visrep_nchw_to_tokens(const void* x, int dtype, void* y, int B, int C, int H, int W, int Cpad, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C || (Cpad & 1)) return visrep_set_error(VISREP_ERR_SHAPE, "nchw_to_tokens: bad shape");
    const long total = (long)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VISREP_F32)
        hipLaunchKernelGGL(nchw_to_tokens<true>, dim3(blocks_for(total)), dim3(256), 0, st, x, (bf16_t*)y, total, C, H * W, Cpad);
    else if (dtype == VISREP_BF16)
        hipLaunchKernelGGL(nchw_to_tokens<false>, dim3(blocks_for(total)), dim3(256), 0, st, x, (bf16_t*)y, total, C, H * W, Cpad);
    else return visrep_set_error(VISREP_ERR_ARG, "nchw_to_tokens: dtype must be bf16 (0) or f32 (1)");
    return launched("nchw_to_tokens: launch failed");
}

extern "C" int visrep_resize_bilinear(const void* x, int dtype, void* y, int planes, int H, int W, int OH, int OW, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "resize_bilinear: empty problem");
    const long total = (long)planes * OH * OW;
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VISREP_F32)
        hipLaunchKernelGGL(resize_bilinear<true>, dim3(blocks_for(total)), dim3(256), 0, st, x, (bf16_t*)y, total, H, W, OH, OW, sh, sw);
    else if (dtype == VISREP_BF16)
        hipLaunchKernelGGL(resize_bilinear<false>, dim3(blocks_for(total)), dim3(256), 0, st, x, (bf16_t*)y, total, H, W, OH, OW, sh, sw);
    else return visrep_set_error(VISREP_ERR_ARG, "resize_bilinear: dtype must be bf16 (0) or f32 (1)");
    return launched("resize_bilinear: launch failed");
}

extern "C" int visrep_sd_noisy_latents(const float* moments, int ldm, const float* post_noise, const float* ddim_noise, void* y, int B,
                                       int Z, int HW, int Cpad, float scaling, float coef_latent, float coef_noise, void* stream) {
    if (B <= 0 || Z <= 0 || HW <= 0 || Cpad < Z || (Cpad & 1) || ldm < 2 * Z) return visrep_set_error(VISREP_ERR_SHAPE, "sd_noisy_latents: bad shape");
    const long total = (long)B * HW;
    hipLaunchKernelGGL(sd_noisy_latents, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, moments, ldm, post_noise, ddim_noise,
                       (bf16_t*)y, total, Z, HW, Cpad, scaling, coef_latent, coef_noise);
    return launched("sd_noisy_latents: launch failed");
}

extern "C" int visrep_mean_groups(const void* x, void* y, int B, int E, long N, void* stream) {
    if (B <= 0 || E <= 0 || N <= 0 || (N & 1)) return visrep_set_error(VISREP_ERR_SHAPE, "mean_groups: N must be even");
    const long total = (long)B * (N / 2);
    hipLaunchKernelGGL(mean_groups, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, total, N / 2, E);
    return launched("mean_groups: launch failed");
}

// ================================================================================================ input pipeline
// PIL-exact separable resampling of 8-bit images (Pillow src/libImaging/Resample.c, 8bpc path): fixed-point coefficients
// (22 fractional bits) prepared on the host exactly like precompute_coeffs + normalize_coeffs_8bpc, int32 accumulation from
// 1 << 21, arithmetic shift, clamp to [0, 255].  One pass = one axis; `line` / `elem` strides select rows or columns.
namespace {
__global__ __launch_bounds__(256) void resample_u8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long total, int out_len,
                                                   int channels, long ils, long ies, long ols, long oes, const int* __restrict__ bounds,
                                                   const int* __restrict__ kk, int ksize) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // (line, xx, c)
    if (idx >= total) return;
    const int c = (int)(idx % channels);
    const long r = idx / channels;
    const int xx = (int)(r % out_len);
    const long line = r / out_len;
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ksize;
    const uint8_t* src = in + line * ils + (long)xmin * ies + c;
    int ss = 1 << 21;
    for (int x = 0; x < cnt; ++x) ss += (int)src[x * ies] * k[x];
    ss >>= 22;
    out[line * ols + (long)xx * oes + c] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

// crop + ToTensor + normalise: out[c, y, x] = (in[y0 + y, x0 + x, c] / 255 - mean[c]) / std[c]  (IEEE fp32 division: bit-identical
// to numpy's `(a / 255.0 - mean) / std` of the CPU processors), fp32 or bf16 output, NCHW
template <bool F32>
__global__ __launch_bounds__(256) void u8hwc_to_chw_norm(const uint8_t* __restrict__ in, void* __restrict__ out, long total, int W, int x0, int y0,
                                                         int S_h, int S_w, float m0, float m1, float m2, float s0, float s1, float s2) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // (c, y, x)
    if (idx >= total) return;
    const int x = (int)(idx % S_w);
    const long r = idx / S_w;
    const int y = (int)(r % S_h), c = (int)(r / S_h);
    const float v = (float)in[((long)(y0 + y) * W + (x0 + x)) * 3 + c] / 255.0f;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const float o = (v - mean) / sd;
    if (F32) reinterpret_cast<float*>(out)[idx] = o;
    else reinterpret_cast<bf16_t*>(out)[idx] = (bf16_t)(pack_bf16(o, 0.f) & 0xffffu);
}
}  // namespace

// ---- the same two passes + crop + normalisation for a whole batch of images of DIFFERENT sizes in two launches (round 3): one int64
// descriptor per image (PD_*) names the source image, an optional virtual canvas around it (llava/mm_utils.py:78-91 expand2square: the
// image pasted centred on a square of the background colour), an optional mirror (Image.FLIP_LEFT_RIGHT), the two coefficient tables
// and the crop.  Pass 1 resamples horizontally only the canvas rows the vertical pass will touch and only the columns inside the crop;
// pass 2 resamples vertically, crops and normalises straight into out[img].  Same integer arithmetic as resample_u8, same float
// expression as u8hwc_to_chw_norm: bit-identical to the per-image route.
namespace {
enum { PD_SRC = 0, PD_H, PD_W, PD_VH, PD_VW, PD_PADY, PD_PADX, PD_BG, PD_FLIP, PD_MID, PD_R0, PD_NR, PD_HB, PD_HK, PD_HKS, PD_VB, PD_VK, PD_VKS,
       PD_X0, PD_Y0, PD_FIELDS = 24 };

struct PdImage {
    const uint8_t* src; long H, W, pady, padx; unsigned bg; bool flip;
    VR_DEV explicit PdImage(const long long* D)
        : src(reinterpret_cast<const uint8_t*>(D[PD_SRC])), H(D[PD_H]), W(D[PD_W]), pady(D[PD_PADY]), padx(D[PD_PADX]), bg((unsigned)D[PD_BG]), flip(D[PD_FLIP] != 0) {}
    VR_DEV int px(long y, long x, int c) const {                // pixel (y, x) of the canvas: the (mirrored) image or the background
        const long ry = y - pady;
        long rx = x - padx;
        if (ry < 0 || ry >= H || rx < 0 || rx >= W) return (int)((bg >> (8 * c)) & 255u);
        if (flip) rx = W - 1 - rx;
        return (int)src[(ry * W + rx) * 3 + c];
    }
};

__global__ __launch_bounds__(256) void preprocess_h_pass(const long long* __restrict__ desc, int crop_w) {
    const long long* D = desc + (long)blockIdx.y * PD_FIELDS;
    const int ks = (int)D[PD_HKS];
    if (ks == 0) return;                                        // canvas width == resized width: pass 2 reads the canvas itself
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // (row - R0, x inside the crop)
    if (idx >= D[PD_NR] * crop_w) return;
    const int x = (int)(idx % crop_w);
    const long rr = idx / crop_w;
    const int xx = (int)D[PD_X0] + x;
    const int* bounds = reinterpret_cast<const int*>(D[PD_HB]);
    const int* k = reinterpret_cast<const int*>(D[PD_HK]) + (long)xx * ks;
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const PdImage im(D);
    const long row = D[PD_R0] + rr;
    uint8_t* mid = reinterpret_cast<uint8_t*>(D[PD_MID]) + idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int ss = 1 << 21;
        for (int i = 0; i < cnt; ++i) ss += im.px(row, xmin + i, c) * k[i];
        ss >>= 22;
        mid[c] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

template <bool F32>
__global__ __launch_bounds__(256) void preprocess_v_pass(const long long* __restrict__ desc, void* __restrict__ out, int crop_h, int crop_w, float m0,
                                                         float m1, float m2, float s0, float s1, float s2) {
    const long long* D = desc + (long)blockIdx.y * PD_FIELDS;
    const int idx = blockIdx.x * 256 + threadIdx.x;             // (y, x) inside the crop
    if (idx >= crop_h * crop_w) return;
    const int x = idx % crop_w, y = idx / crop_w;
    const int hks = (int)D[PD_HKS], vks = (int)D[PD_VKS];
    const int yy = (int)D[PD_Y0] + y, xx = (int)D[PD_X0] + x;
    const PdImage im(D);
    const uint8_t* mid = reinterpret_cast<const uint8_t*>(D[PD_MID]);
    const long r0 = D[PD_R0];
    auto fetch = [&](long row, int c) -> int { return hks ? (int)mid[((row - r0) * crop_w + x) * 3 + c] : im.px(row, xx, c); };
    int ymin = yy, cnt = 1;
    const int* k = nullptr;
    if (vks) {
        const int* bounds = reinterpret_cast<const int*>(D[PD_VB]);
        ymin = bounds[2 * yy];
        cnt = bounds[2 * yy + 1];
        k = reinterpret_cast<const int*>(D[PD_VK]) + (long)yy * vks;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int u;
        if (vks) {
            int ss = 1 << 21;
            for (int i = 0; i < cnt; ++i) ss += fetch(ymin + i, c) * k[i];
            ss >>= 22;
            u = ss < 0 ? 0 : (ss > 255 ? 255 : ss);
        } else {
            u = fetch(yy, c);
        }
        const float v = (float)u / 255.0f;
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        const float o = (v - mean) / sd;
        const long oi = (((long)blockIdx.y * 3 + c) * crop_h + y) * crop_w + x;
        if (F32) reinterpret_cast<float*>(out)[oi] = o;
        else reinterpret_cast<bf16_t*>(out)[oi] = (bf16_t)(pack_bf16(o, 0.f) & 0xffffu);
    }
}
}  // namespace

extern "C" int visrep_preprocess_u8_batch(const void* desc, int n_images, long max_mid_rows, int crop_h, int crop_w, const float* mean3,
                                          const float* std3, void* out, int dtype, void* stream) {
    if (n_images <= 0) return 0;
    if (!desc || !out || !mean3 || !std3) return visrep_set_error(VISREP_ERR_ARG, "preprocess_u8_batch: null pointer (mean3 / std3 are HOST arrays)");
    if (crop_h <= 0 || crop_w <= 0 || max_mid_rows < 0) return visrep_set_error(VISREP_ERR_SHAPE, "preprocess_u8_batch: empty crop");
    if (dtype != VISREP_F32 && dtype != VISREP_BF16) return visrep_set_error(VISREP_ERR_ARG, "preprocess_u8_batch: dtype must be bf16 (0) or f32 (1)");
    hipStream_t st = (hipStream_t)stream;
    const long long* d = reinterpret_cast<const long long*>(desc);
    if (max_mid_rows > 0)
        hipLaunchKernelGGL(preprocess_h_pass, dim3((unsigned)((max_mid_rows * crop_w + 255) / 256), n_images), dim3(256), 0, st, d, crop_w);
    const dim3 grid((unsigned)(((long)crop_h * crop_w + 255) / 256), n_images);
    if (dtype == VISREP_F32)
        hipLaunchKernelGGL(preprocess_v_pass<true>, grid, dim3(256), 0, st, d, out, crop_h, crop_w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    else
        hipLaunchKernelGGL(preprocess_v_pass<false>, grid, dim3(256), 0, st, d, out, crop_h, crop_w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return launched("preprocess_u8_batch: launch failed");
}

extern "C" int visrep_resample_u8(const void* in, void* out, long n_lines, int out_len, int channels, long in_line_stride,
                                  long in_elem_stride, long out_line_stride, long out_elem_stride, const int* bounds, const int* kk, int ksize,
                                  void* stream) {
    if (!in || !out || !bounds || !kk) return visrep_set_error(VISREP_ERR_ARG, "resample_u8: null pointer");
    if (n_lines <= 0 || out_len <= 0 || channels <= 0 || ksize <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "resample_u8: empty problem");
    const long total = n_lines * out_len * channels;
    hipLaunchKernelGGL(resample_u8, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)in, (uint8_t*)out, total, out_len,
                       channels, in_line_stride, in_elem_stride, out_line_stride, out_elem_stride, bounds, kk, ksize);
    return launched("resample_u8: launch failed");
}

extern "C" int visrep_u8hwc_to_chw_norm(const void* in, int H, int W, int x0, int y0, int crop_h, int crop_w, const float* mean3,
                                        const float* std3, void* out, int dtype, void* stream) {
    if (!in || !out || !mean3 || !std3) return visrep_set_error(VISREP_ERR_ARG, "u8hwc_to_chw_norm: null pointer (mean3 / std3 are HOST arrays)");
    if (x0 < 0 || y0 < 0 || crop_h <= 0 || crop_w <= 0 || x0 + crop_w > W || y0 + crop_h > H)
        return visrep_set_error(VISREP_ERR_SHAPE, "u8hwc_to_chw_norm: crop outside the image");
    const long total = 3L * crop_h * crop_w;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VISREP_F32)
        hipLaunchKernelGGL(u8hwc_to_chw_norm<true>, dim3(blocks_for(total)), dim3(256), 0, st, (const uint8_t*)in, out, total, W, x0, y0, crop_h,
                           crop_w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    else if (dtype == VISREP_BF16)
        hipLaunchKernelGGL(u8hwc_to_chw_norm<false>, dim3(blocks_for(total)), dim3(256), 0, st, (const uint8_t*)in, out, total, W, x0, y0, crop_h,
                           crop_w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    else return visrep_set_error(VISREP_ERR_ARG, "u8hwc_to_chw_norm: dtype must be bf16 (0) or f32 (1)");
    return launched("u8hwc_to_chw_norm: launch failed");
}
