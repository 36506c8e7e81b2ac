// First convolution of the diffusion VAE encoder (diffusers vae.py Encoder.conv_in: Conv2d(3, 128, 3, padding = 1) on the 768^2 / 512^2 image;
// dift_sd.py:172 vae.encode), gfx950.
//
// Until round 5 this layer ran as im2col (3 channels padded to 8: a [B H W, 128] bf16 matrix, 2.4 GB written and read back at 16 x 768^2) + the
// 128x128 GEMM, and the GroupNorm that follows read the 2.4-GB output once more for its statistics: 1.7 ms of the 70-ms SD1.5 forward for 65 GFLOP.
// The input is tiny (16 B per pixel as an 8-channel token) and the output is the traffic, so: no staging at all.
//   * D^T = W X^T with 16x16x32 MFMAs: the B operand of k-step ks is, for lane (pixel p = lane & 15, k-group kg = lane >> 4), the 8 channels of tap
//     t = 4 ks + kg of pixel p - exactly ONE 16-byte token of the neighbour pixel (y + ky - 1, x + kx - 1): one global load per lane and k-step,
//     zeros outside the image and for the three padding taps 9 .. 11 (K = 9 x 8 = 72 -> 96);
//   * the A operand is the weight matrix [128, 96] in the packer's K order (tap, channel): 16-byte loads, L1-resident (24 KB);
//   * a wave owns 64 consecutive pixels x 64 output channels (two waves share a pixel group: 128 accumulator registers per wave spilled); accumulators have the GEMM kernels' geometry (four consecutive channels of one
//     pixel per lane), so the epilogue is theirs: bias, 16-byte stores through permlane swaps, and the GroupNorm partial sums of the OUTPUT per
//     64-pixel slot and group (gemm_epilogue.h HAS_GN) - the first ResnetBlock2D's norm1 needs no pass over the tensor.
// Scope: 8-channel tokens in, Cout = 128, stride 1, padding 1, W % 16 == 0, H W % 64 == 0.  Results equal im2col + GEMM up to the order of the
// fp32 sum inside an MFMA (tests/test_gpu_sd.py).
#include "common.h"
#include "gemm_epilogue.h"
#include "visrep_internal.h"

namespace {

struct ConvInArgs {
    const bf16_t* x;        // [B H W, 8] bf16 tokens (visrep_nchw_to_tokens with Cpad = 8)
    const bf16_t* w;        // [128, ldw >= 96], K order (ky, kx, c8); columns 72 .. 95 zero
    int B, H, W, ldw;
    GemmArgs g;             // C, ldc, bias, M, N = 128, gn_partial / gn_cpg / gn_hw: what the shared epilogue reads
};

__global__ __launch_bounds__(256, 3) void conv3x3_c8(const ConvInArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, kg = lane >> 4;
    const long base = ((long)blockIdx.x * 2 + (wave >> 1)) * 64;               // first pixel of this wave's group (row-major over B, H, W)
    const int nb = (wave & 1) * 64;                                            // its half of the output channels
    if (base >= p.g.M) return;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // (image, row, first column) of the wave's four 16-pixel blocks: one 32-bit division pair per wave (uniform), then steps of 16 columns -
    // a block lies inside one image row (W % 16 == 0); 64-bit divisions per block cost this kernel a third of its time
    int by[4], bx[4];                                                          // row index over all images (b H + y), first column
    {
        const unsigned ub = (unsigned)base, hw = (unsigned)(p.H * p.W);
        const unsigned b = ub / hw, r = ub - b * hw, y = r / (unsigned)p.W;
        int gy = (int)(b * (unsigned)p.H + y), x = (int)(r - y * (unsigned)p.W);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            by[i] = gy; bx[i] = x;
            x += 16;
            if (x >= p.W) { x = 0; ++gy; }
        }
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const int tap = ks * 4 + kg, ky = tap / 3, kx = tap - ky * 3;         // taps 9 .. 11: padding of K
        bf16x8 xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yimg = by[i] % p.H;                                      // row inside its image (uniform)
            const int yy = yimg + ky - 1, xx = bx[i] + px + kx - 1;
            const bool in = tap < 9 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const bf16_t* src = p.x + ((long)(by[i] + (in ? ky - 1 : 0)) * p.W + (in ? xx : bx[i] + px)) * 8;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
            xf[i] = in ? v : zero;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(nb + j * 16 + px) * p.ldw + ks * 32 + kg * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[i], acc[i][j], 0, 0, 0);
        }
    }
    const int mb = (int)base;
    if (p.g.gn_partial) gemm_epilogue_rowmajor_gn<EPI_BIAS, 4, 4>(p.g, acc, mb, nb, px, kg);
    else gemm_epilogue_rowmajor<EPI_BIAS, 4, 4>(p.g, acc, mb, nb, px, kg);
}

}  // namespace

extern "C" int visrep_conv3x3_c8_supported(int B, int H, int W, int Cout) {
    return B > 0 && H > 0 && W > 0 && Cout == 128 && W % 16 == 0 && ((long)H * W) % 64 == 0 && (long)B * H * W < (1L << 31) - 64;
}

extern "C" int visrep_conv3x3_c8_bf16(const void* x, int B, int H, int W, const void* Wt, int ldw, const float* bias, void* out, int ldc, int Cout,
                                      void* gn_partial, int groups, void* stream) {
    if (!x || !Wt || !out) return visrep_set_error(VISREP_ERR_ARG, "conv3x3_c8: null pointer");
    if (!visrep_conv3x3_c8_supported(B, H, W, Cout)) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_c8: needs Cout = 128, W % 16 == 0, H W % 64 == 0");
    if (ldw < 96 || (ldw & 7) || ldc < Cout || (ldc & 7)) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_c8: ldw >= 96, ldc >= Cout, both multiples of 8");
    if (gn_partial && (groups <= 0 || Cout % groups || (Cout / groups != 4 && Cout / groups != 8 && Cout / groups != 16) || ((long)H * W) % 128))
        return visrep_set_error(VISREP_ERR_ARG, "conv3x3_c8: GroupNorm partials need 4 | 8 | 16 channels per group and H W % 128 == 0");
    ConvInArgs a{};
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)Wt; a.B = B; a.H = H; a.W = W; a.ldw = ldw;
    a.g.C = (bf16_t*)out; a.g.ldc = ldc; a.g.bias = bias; a.g.M = B * H * W; a.g.N = Cout; a.g.K = 96; a.g.epi = EPI_BIAS;
    if (gn_partial) { a.g.gn_partial = (float2*)gn_partial; a.g.gn_cpg = Cout / groups; a.g.gn_hw = H * W; }
    visrep_count_route(VISREP_ROUTE_CONV_C8);
    const long groups64 = ((long)a.g.M + 63) / 64;                             // two pixel groups x two channel halves per workgroup
    hipLaunchKernelGGL(conv3x3_c8, dim3((unsigned)((groups64 + 1) / 2)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "conv3x3_c8: launch failed");
}
