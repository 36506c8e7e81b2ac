// 3x3 convolution with the input's GroupNorm + SiLU fused into the operand path - the VAE encoder's 128-channel layers at 768^2 / 384^2
// (diffusers resnet.py ResnetBlock2D: norm -> SiLU -> conv, vae.py Encoder; dift_sd.py:172 vae.encode), gfx950.
//
// Why a kernel of its own (profiles/round5_sd15_kernel_stats.md).  The implicit-GEMM convolutions (gemm_bf16.hip, gemm_bf16_v5.hip) gather
// their A operand with LDS-DMA: nothing can be computed on the way, so GroupNorm(+SiLU) has to be APPLIED by a separate streaming pass that
// reads and re-writes the tensor (2.4 GB each way at 768^2 x 128 channels x 16 images: 0.87 ms next to a 2.7-ms convolution), and every
// output tile re-gathers its 3x3 neighbourhood nine times (1.18 MB of L2 -> LDS traffic per 256 output pixels).  Here a workgroup owns a
// 16 x 16 pixel output tile of one image:
//   * its 18 x 18 x 128-channel input HALO is fetched ONCE (81 KB), normalised + SiLU'd in registers with the (scale, shift) pair of its
//     (image, channel) - the arithmetic of groupnorm_apply_rows, bit for bit - and parked in LDS; out-of-image pixels are zeros, as
//     Conv2d's padding pads the NORMALISED tensor;
//   * the nine taps are nine shifted views of that halo: an A fragment of tap (ky, kx) is a ds_read_b128 at pixel (y + ky, x + kx).  A pixel
//     is one 256-byte LDS row of sixteen 16-byte channel chunks; chunk c sits in slot c ^ f(hx), f(hx) = 3 (hx & 1) | (hx & 6) << 1 - found
//     by exhaustive search: every 16-lane service group of ds_read_b128 then touches sixteen distinct slots for EVERY tap shift;
//   * only W streams: one 64-deep K-tile (Cout x 128 B) per step through an LDS-DMA ring, one barrier per K-tile, counted vmcnt waits;
//   * 8 waves, 4 (M) x 2 (N), each 64 pixels x Cout / 2 channels in 16x16x32 MFMAs; K order (tap, channel half) and the accumulator chains
//     are those of the 128x128 kernel, so conv(GN(x)) here == conv128(groupnorm_apply(x)) BIT FOR BIT (tests/test_gpu_sd.py);
//   * epilogue: bias (+ residual), 16-byte stores, and optionally the GroupNorm partial sums of the OUTPUT (per 64-pixel wave slot and
//     group) for the norm that follows - so a ResnetBlock2D is two launches + two tiny table kernels, no apply pass, no statistics pass.
// Scope: Cin = 128, Cout in {128, 256}, stride 1, padding 1, H and W multiples of 16.  Everything else keeps the implicit-GEMM kernels.
#include <cstdlib>
#include "common.h"
#include "gemm_epilogue.h"
#include "visrep_internal.h"

// 1: fetch and normalise the NEXT tile's halo under this tile's last K-tiles (registers -> LDS behind the K loop's closing barrier).  Built and
// measured in round 5: the 44 registers of the staged halo on top of 64 accumulators + double-buffered fragments spill (45 registers, reloads
// inside the K loop with the vmcnt(0) the compiler puts behind them) - 3.89 ms against 2.88 ms for the simple order at 768^2 x 16; kept as a knob.
#ifndef HALO_PREFETCH
#define HALO_PREFETCH 0
#endif

namespace {

constexpr int HT = 16, HP = HT + 2, HC = 128;                   // tile width (and the full tile's height), halo width, input channels
constexpr int halo_pixels(int ty) { return (ty + 2) * HP; }     // TY output rows: 16 (one workgroup of 8 waves per CU) or 8 (4 waves, two workgroups per CU)
constexpr int halo_bytes(int ty) { return halo_pixels(ty) * HC * 2; }      // 82,944 / 46,080
constexpr int HROW = HP * 256;                                  // bytes per halo pixel row (18 pixels x 256 B)
constexpr int NKT = 18;                                         // K-tiles: 9 taps x 2 channel halves of 64

struct HaloArgs {
    const bf16_t* x;        // [B, H, W, 128] bf16 channels-last
    const bf16_t* w;        // [Cout, ldw >= 1152], K order (ky, kx, c)
    const float* bias;      // [Cout] or null
    const bf16_t* resid;    // [B H W, ldc] or null (RESID)
    bf16_t* out;            // [B H W, ldc]
    const float2* gn_tab;   // [B, 128] (scale, shift) of the INPUT's GroupNorm, or null = the input is used as it is
    float2* gn_partial;     // GroupNorm partial sums of the OUTPUT [B][H W / 64][groups] or null
    int B, H, W, ldw, ldc, silu, gn_cpg;
    int tiles_x, tiles_per_img, ntiles;
};

VR_DEV int halo_swz(int hx) { return ((hx & 1) * 3) | ((hx & 6) << 1); }
VR_DEV unsigned lds_off(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

// fragment reads as ONE asm statement each (the compiler must not see LDS reads next to in-flight LDS-DMA: it would drain the W ring with
// a vmcnt(0) per K-tile - gemm_bf16_v5.hip); A: four pixel rows of the tile (4608 B apart), W: NJ 16-row blocks (2048 B apart)
template <int OFF> VR_DEV void issue_a4(bf16x8 (&a)[4], unsigned addr) {      // OFF = ky * HROW: the tap's row shift rides in the immediate offsets
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]) : "v"(addr), "n"(OFF), "n"(OFF + 4608), "n"(OFF + 9216), "n"(OFF + 13824));
}
VR_DEV void issue_w4(bf16x8 (&w)[4], unsigned addr) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(addr));
}
VR_DEV void issue_w8(bf16x8 (&w)[8], unsigned addr) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:2048\n\tds_read_b128 %2, %8 offset:4096\n\tds_read_b128 %3, %8 offset:6144\n\t"
                 "ds_read_b128 %4, %8 offset:8192\n\tds_read_b128 %5, %8 offset:10240\n\tds_read_b128 %6, %8 offset:12288\n\tds_read_b128 %7, %8 offset:14336"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]) : "v"(addr));
}
template <int NJ> VR_DEV void wait_frags(bf16x8 (&a)[4], bf16x8 (&w)[NJ]) {
    if constexpr (NJ == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]),
                     "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
}
// counted form: all but the newest N LDS reads of this wave have landed; the named fragments may be consumed behind it
template <int NJ, int N> VR_DEV void wait_frags_n(bf16x8 (&a)[4], bf16x8 (&w)[NJ]) {
    if constexpr (NJ == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N));
    else
        asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]),
                     "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "n"(N));
}
template <int N> VR_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
VR_DEV void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// NJ: 16-column blocks per wave (4: Cout 128, 8: Cout 256); NW: W ring slots; TY: output rows per tile.  TY = 16: 8 waves as 4 (M) x 2 (N), 147 KB of
// LDS, one workgroup per CU - its halo phase runs with the matrix pipe idle (the default).  TY = 8 (round 5, second half): 4 waves as 2 x 2, a 10-row halo
// (46 KB) + a two-slot W ring (32 KB) = 77 KB: TWO workgroups per CU, one's halo phase under the other's K loop, for 11 % more halo bytes per output pixel -
// built to hide the halo phase, measured 3 % slower (2.99 against 2.90 ms), kept as a tested variant (visrep_set_conv_halo_tile).
template <int NJ, int NW, bool RESID, int TY = 16>
__global__ __launch_bounds__((TY / 4) * 128, 2) void conv3x3_halo(const HaloArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = (TY / 4) * 128, HPIX = halo_pixels(TY), HALO_BYTES = halo_bytes(TY);
    constexpr int COUT = NJ * 32, WT = COUT * 128, P = COUT * 8 / NT, D = NW - 1;
    static_assert(D >= 1 && P * D <= 8, "ring depth");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, hg = lane >> 4;
    char* const halo = smem;
    char* const wring = smem + HALO_BYTES;

    // ---- W staging: piece j covers chunk c = j * NT + tid: row c >> 3, physical slot c & 7 <- logical slot (c & 7) ^ ((row >> 1) & 7)
    unsigned wofs[P];                                            // element offset of this lane's chunk inside W (32 bits: scalar base + vector offset addressing)
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int c = j * NT + tid, row = c >> 3;
        wofs[j] = (unsigned)row * (unsigned)p.ldw + (unsigned)((((c & 7) ^ ((row >> 1) & 7))) << 3);
    }
    int g_issue = 0;                                             // stream index of the next W tile to issue
    auto issue_w = [&](int kt_issue) {                           // kt_issue: compile-time at every call site (K offset = an immediate)
        char* dst = wring + (g_issue % NW) * WT + wave * 1024;
#pragma unroll
        for (int j = 0; j < P; ++j) glds16(p.w + (wofs[j] + (unsigned)(kt_issue * 64)), dst + j * (NT * 16));
        ++g_issue;
    };

    // ---- fragment addresses.  A: lane (fr, hg) reads pixel (row, hx = fr + kx), chunk cu + hg (cu = 8 h + 4 kk): slot (cu ^ (f & 12)) | (hg ^ (f & 3))
    // twelve per-lane addresses cover every (kx, channel quarter); ky and the accumulator row i are immediate offsets of the reads
    unsigned aaddr[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int hx = fr + kx, f = halo_swz(hx);
        const unsigned ab = lds_off(halo) + (unsigned)(wm * 4 * HROW + hx * 256 + ((hg ^ (f & 3)) << 4)), ah = (unsigned)((f & 12) << 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) aaddr[kx][q] = ab + ((unsigned)(q << 6) ^ ah);       // q = 2 h + kk: chunk base 4 q
    }
    // W: row wn * COUT / 2 + 16 j + fr, logical slot 4 kk + hg -> physical ((4 kk) ^ (s & 4)) | (hg ^ (s & 3)), s = (fr >> 1) & 7
    const int s3 = (fr >> 1) & 7;
    const unsigned wfrag = (unsigned)((wn * (COUT / 2) + fr) * 128 + ((hg ^ (s3 & 3)) << 4)), whi = (unsigned)((s3 & 4) << 4);
    const unsigned wring_off = lds_off(wring);
    const unsigned wadr0 = wring_off + wfrag + (0u ^ whi), wadr1 = wring_off + wfrag + (64u ^ whi);      // k-step 0 / 1 of slot 0

    // ---- halo fill: task q = it * NT + tid -> pixel q >> 4, chunk q & 15 = tid & 15 (fixed per thread: its (scale, shift) octet is loaded once per tile)
    const int chunk = tid & 15;

    // prologue of the W stream
#pragma unroll
    for (int d = 0; d < D; ++d) issue_w(d);
    int g = 0;                                                   // stream index of the K-tile being consumed

    const int G = gridDim.x;
    const int lid = (blockIdx.x & 7) * ((G + 7) >> 3) + (blockIdx.x >> 3);      // blocks of one XCD (b % 8) walk neighbouring tiles: shared halo rows in L2
    const int tstride = ((G + 7) >> 3) * 8;
    constexpr int NIT = (HPIX * 16 + NT - 1) / NT;              // 11 (TY = 16) / 12 (TY = 8) sixteen-byte tasks per thread and halo
    // ---- the halo of a tile in three steps, so that the NEXT tile's halo can be fetched and normalised under this tile's MFMAs (PREFETCH, the
    // Cout = 128 variant): load (global -> registers), transform (GroupNorm + SiLU in registers), store (registers -> LDS, after the barrier
    // that ends the readers of the current halo).  sc / sh always belong to the halo that is transformed next.
    constexpr bool PREFETCH = HALO_PREFETCH && NJ == 4;
    float sc[8], sh[8];
    u32x4 raw[NIT];
    unsigned inmask = 0;                                         // bit it: task it lies inside the image (padding pixels stay zero)
    auto load_tab = [&](int b) {
        if (p.gn_tab) {
            const float4* tb = reinterpret_cast<const float4*>(p.gn_tab + (size_t)b * HC + chunk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float4 v = tb[e]; sc[2 * e] = v.x; sh[2 * e] = v.y; sc[2 * e + 1] = v.z; sh[2 * e + 1] = v.w; }
        }
    };
    auto halo_load = [&](int t) {                                // branch-free: a task outside the image loads a clamped (valid) address and is zeroed
        const int b = t / p.tiles_per_img, r = t - b * p.tiles_per_img;    // by the transform - a divergent `if (inside) load` costs a vmcnt(0) per task
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x, y0 = ty * TY, x0 = tx * HT;
        const bf16_t* img = p.x + (size_t)b * p.H * p.W * HC + chunk * 8;
        inmask = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = it * NT + tid, pix = q >> 4;
            const int hy = pix / HP, hx = pix - hy * HP;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool in = pix < HPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            inmask |= (unsigned)in << it;
            const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
            raw[it] = *reinterpret_cast<const u32x4*>(img + (size_t)(cy * p.W + cx) * HC);
        }
    };
    auto halo_xform = [&](auto IT) __attribute__((always_inline)) {       // groupnorm_apply_rows' arithmetic, bit for bit; padding pixels become zeros
        constexpr int it = decltype(IT)::value;
        const bool in = (inmask >> it) & 1u;
        u32x4 o4 = raw[it];
        if (p.gn_tab) {                                          // uniform
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(raw[it][e]); v[2 * e + 1] = bf_hi(raw[it][e]); }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = __builtin_fmaf(v[e], sc[e], sh[e]);
                if (p.silu) o = o * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(o * -1.4426950408889634f));
                v[e] = o;
            }
            o4 = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
        }
        raw[it] = u32x4{in ? o4[0] : 0u, in ? o4[1] : 0u, in ? o4[2] : 0u, in ? o4[3] : 0u};
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = it * NT + tid, pix = q >> 4;
            if (pix >= HPIX) continue;
            const int hx = pix - (pix / HP) * HP;
            *reinterpret_cast<u32x4*>(halo + pix * 256 + ((chunk ^ halo_swz(hx)) << 4)) = raw[it];
        }
    };
    bool staged = false;                                         // raw[] holds this tile's halo, normalised, waiting for its LDS store
    for (int t = lid; t < p.ntiles; t += tstride) {
        const int b = t / p.tiles_per_img, r = t - b * p.tiles_per_img;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x, y0 = ty * TY, x0 = tx * HT;
        // ------------------------------------------------------------ halo
        if (!staged) {                                           // first tile of the block (or no prefetch): fetch and normalise here
            halo_load(t);                                        // eleven 16-byte loads in flight, then the (scale, shift) octet behind them
            load_tab(b);
            static_for<NIT>(halo_xform);
            __syncthreads();                                     // every wave is done reading the previous tile's halo
            halo_store();
        }
        __syncthreads();                                         // the halo is in LDS (prefetched tiles: stored behind the previous K loop's closing barrier)
        const int tn = t + tstride;
        const bool has_next = PREFETCH && tn < p.ntiles;

        // ------------------------------------------------------------ K loop: 9 taps x 2 channel halves, W through the ring.  Fully unrolled (the tap
        // geometry is compile-time), fragments double-buffered: the reads of k-step kk + 1 are in flight under the MFMAs of k-step kk, and the A
        // fragments of the NEXT K-tile (the halo never changes during a tile: no barrier needed) under the MFMAs of this tile's second k-step.
        // LDS returns in order, so the waits are counted (lgkmcnt(n) = all but the newest n reads have landed).
        f32x4 acc[4][NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr bool PIPE = NJ == 4;                           // double-buffered fragments (the Cout = 256 variant has no registers for a second set)
        bf16x8 xa[PIPE ? 2 : 1][4], xw[PIPE ? 2 : 1][NJ];
        // the W source offsets are re-"defined" once per tile: otherwise hipcc hoists the 18 x P 64-bit source pointers of the unrolled K loop out
        // of the tile loop and spills them (72 registers); two 64-bit adds per K-tile are cheaper than their reloads
#pragma unroll
        for (int j = 0; j < P; ++j) asm volatile("" : "+v"(wofs[j]));
        if constexpr (PIPE) issue_a4<0>(xa[0], aaddr[0][0]);
        static_for<NKT>([&](auto KT) __attribute__((always_inline)) {
            constexpr int kt = decltype(KT)::value, tp = kt >> 1, h = kt & 1, ky = tp / 3, kx = tp - 3 * ky;
            // this wave's pieces of W(g) have landed; newer tiles' pieces - and, for three K-tiles, the next halo's loads issued behind W(15) - may fly
            if constexpr (PREFETCH && kt >= 13 && kt <= 15) { if (has_next) wait_vm<P * (D - 1) + NIT>(); else wait_vm<P * (D - 1)>(); }
            else wait_vm<P * (D - 1)>();
            wg_barrier();                                        // ... everybody's have, and everybody is done with the slot W(g + D) will take
            issue_w((kt + D) % NKT);
            if constexpr (PREFETCH && kt == 12) {                // the next tile's halo: 11 loads per lane, consumed four K-tiles later
                __builtin_amdgcn_sched_barrier(0);               // pinned: the counted waits below assume exactly this position in the VMEM queue
                if (has_next) halo_load(tn);
                __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned so = (unsigned)((g % NW) * WT);
            if constexpr (PIPE) {
                issue_w4(xw[0], wadr0 + so);
                issue_a4<ky * HROW>(xa[1], aaddr[kx][2 * h + 1]);
                issue_w4(xw[1], wadr1 + so);
                wait_frags_n<NJ, 4 + NJ>(xa[0], xw[0]);          // A(kt, 0) and W(kt, 0) are in; A(kt, 1) / W(kt, 1) still fly
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw[0][j], xa[0][i], acc[i][j], 0, 0, 0);
                if constexpr (kt + 1 < NKT) {
                    __builtin_amdgcn_sched_barrier(0);           // the reload of xa[0] must stay behind the MFMAs that read it
                    constexpr int kn = kt + 1, tpn = kn >> 1, hn = kn & 1, kyn = tpn / 3, kxn = tpn - 3 * kyn;
                    issue_a4<kyn * HROW>(xa[0], aaddr[kxn][2 * hn]);
                    wait_frags_n<NJ, 4>(xa[1], xw[1]);
                } else {
                    wait_frags_n<NJ, 0>(xa[1], xw[1]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw[1][j], xa[1][i], acc[i][j], 0, 0, 0);
            } else {                                             // 128 accumulators: one fragment set, read - wait - multiply per k-step
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    issue_w8(xw[0], (kk ? wadr1 : wadr0) + so);
                    issue_a4<ky * HROW>(xa[0], aaddr[kx][2 * h + kk]);
                    wait_frags_n<NJ, 0>(xa[0], xw[0]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw[0][j], xa[0][i], acc[i][j], 0, 0, 0);
                }
            }
            if constexpr (PREFETCH && kt >= 16) {               // normalise the prefetched halo in registers, under this K-tile's MFMAs
                if (has_next) {
                    if constexpr (kt == 16) {
                        const int bn = tn / p.tiles_per_img;
                        if (bn != b) load_tab(bn);               // the next tile belongs to the next image: its (scale, shift) octet
                        static_for<6>(halo_xform);
                    } else {
                        static_for<NIT - 6>([&](auto I) __attribute__((always_inline)) { halo_xform(std::integral_constant<int, decltype(I)::value + 6>{}); });
                    }
                }
            }
            ++g;
        });
        if constexpr (PREFETCH) {
            __syncthreads();                                     // every wave is done reading this tile's halo ...
            if (has_next) halo_store();                          // ... the next one goes in; it becomes visible at the barrier that opens the next tile
            staged = has_next;
        }

        // ------------------------------------------------------------ epilogue: bias (+ residual) -> bf16, 16-byte stores; GroupNorm partials of the output
        const int nb = wn * (COUT / 2), lane_col = nb + hg * 4, wide_col = lane_col + ((hg & 1) ? 12 : 0);
        float4 bv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = float4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(p.bias + lane_col + j * 16);
        }
        float gs1[NJ], gs2[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { gs1[j] = 0.f; gs2[j] = 0.f; }
        const size_t row0 = ((size_t)b * p.H + y0 + wm * 4) * p.W + x0 + fr;    // output row of accumulator block i: row0 + i W
        static_for<4>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const size_t orow = row0 + (size_t)i * p.W;
            char* crow = reinterpret_cast<char*>(p.out) + (orow * p.ldc + wide_col) * 2;
            u32x4 rq[RESID ? NJ / 2 : 1];
            if (RESID) {
                const bf16_t* rrow = p.resid + orow * p.ldc + wide_col;
#pragma unroll
                for (int c0 = 0; c0 < NJ; c0 += 2) rq[c0 / 2] = *reinterpret_cast<const u32x4*>(rrow + 16 * c0);
            }
            static_for<NJ / 2>([&](auto C0) __attribute__((always_inline)) {
                constexpr int c0 = decltype(C0)::value * 2;
                u32x2 rv[2], o2[2];
                if (RESID) {                                     // even lanes loaded (own j, partner's j), odd lanes (partner's j + 1, own j + 1): swap back
                    const u32x4 q4 = rq[RESID ? c0 / 2 : 0];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(q4[0], q4[2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(q4[1], q4[3], false, false);
                    rv[0] = u32x2{(unsigned)s0[0], (unsigned)s1[0]};
                    rv[1] = u32x2{(unsigned)s0[1], (unsigned)s1[1]};
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int j = c0 + cc;
                    float v0 = acc[i][j][0] + bv[j].x, v1 = acc[i][j][1] + bv[j].y, v2 = acc[i][j][2] + bv[j].z, v3 = acc[i][j][3] + bv[j].w;
                    if (RESID) { v0 += bf_lo(rv[cc][0]); v1 += bf_hi(rv[cc][0]); v2 += bf_lo(rv[cc][1]); v3 += bf_hi(rv[cc][1]); }
                    gs1[j] += (v0 + v1) + (v2 + v3);
                    gs2[j] = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, gs2[j]))));
                    o2[cc] = u32x2{pack_bf16(v0, v1), pack_bf16(v2, v3)};
                }
                const auto w0 = __builtin_amdgcn_permlane16_swap(o2[0][0], o2[1][0], false, false);
                const auto w1 = __builtin_amdgcn_permlane16_swap(o2[0][1], o2[1][1], false, false);
                const u32x4 q4 = {(unsigned)w0[0], (unsigned)w1[0], (unsigned)w0[1], (unsigned)w1[1]};
                store_b128_at<c0 * 16 * 2>(crow, __builtin_bit_cast(f32x4, q4));
            });
        });
        if (p.gn_partial) {                                      // slot = this wave's 64 pixels (tile, wm); groups of this wave's columns
            const int Gn = COUT / p.gn_cpg, nblk = (p.H * p.W) >> 6;
            float2* dst = p.gn_partial + ((size_t)b * nblk + (size_t)r * (TY / 4) + wm) * Gn;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float s1 = sum_over_fr(gs1[j]), s2 = sum_over_fr(gs2[j]);
                if (p.gn_cpg >= 8) {
                    const auto a1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s1), false, false);
                    const auto a2 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s2), __builtin_bit_cast(unsigned, s2), false, false);
                    s1 = __builtin_bit_cast(float, (unsigned)a1[0]) + __builtin_bit_cast(float, (unsigned)a1[1]);
                    s2 = __builtin_bit_cast(float, (unsigned)a2[0]) + __builtin_bit_cast(float, (unsigned)a2[1]);
                }
                if (p.gn_cpg >= 16) {
                    const auto a1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s1), false, false);
                    const auto a2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s2), __builtin_bit_cast(unsigned, s2), false, false);
                    s1 = __builtin_bit_cast(float, (unsigned)a1[0]) + __builtin_bit_cast(float, (unsigned)a1[1]);
                    s2 = __builtin_bit_cast(float, (unsigned)a2[0]) + __builtin_bit_cast(float, (unsigned)a2[1]);
                }
                const int colq = nb + j * 16 + hg * 4;
                if (fr == 0 && (colq % p.gn_cpg) == 0) {
                    const u32x2 o = {__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s2)};
                    store_b64(dst + colq / p.gn_cpg, o);
                }
            }
        }
        drain_visible_loads();                                   // bias / residual loads are retired before the next tile's counted waits
    }
    wait_vm<0>();                                                // the run-ahead W tiles nobody consumes
}

// (scale, shift) per (image, channel) of a GroupNorm whose statistics are known: scale = rstd gamma, shift = beta - mean scale (groupnorm_apply_rows)
__global__ __launch_bounds__(256) void gn_table_kernel(const float2* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float2* __restrict__ tab, int BC, int C, int cpg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BC) return;
    const int b = i / C, c = i - b * C, G = C / cpg;
    const float2 st = stats[(size_t)b * G + c / cpg];
    const float gsc = st.y * gamma[c];
    tab[i] = float2{gsc, __builtin_fmaf(-st.x, gsc, beta[c])};
}

template <int NJ, int NW, bool RESID, int TY = 16>
int launch_halo(HaloArgs a, hipStream_t s) {
    constexpr int LDS = halo_bytes(TY) + NW * NJ * 32 * 128;
    static VisrepLdsOptIn opt;
    visrep_lds_opt_in(opt, reinterpret_cast<const void*>(conv3x3_halo<NJ, NW, RESID, TY>), LDS);
    a.tiles_x = a.W / HT; a.tiles_per_img = (a.H / TY) * (a.W / HT); a.ntiles = a.B * a.tiles_per_img;
    const int cap = visrep_cu_count() * (TY == 16 ? 1 : 2) / 8 * 8;     // resident workgroups: one (147 KB of LDS) or two (77 KB) per CU
    const int want = (a.ntiles + 7) / 8 * 8;                     // a multiple of 8: the XCD-contiguous tile walk is a bijection then (idle blocks just exit)
    hipLaunchKernelGGL((conv3x3_halo<NJ, NW, RESID, TY>), dim3(want < cap ? want : cap), dim3((TY / 4) * 128), LDS, s, a);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "conv3x3_halo: launch failed");
}

}  // namespace

namespace {
std::atomic<int> g_halo_tile{0};                                 // 0: not decided yet (VISREP_HALO_TILE), else 8 | 16
int halo_tile_rows() {
    int v = g_halo_tile.load(std::memory_order_relaxed);
    if (!v) {
        const char* e = getenv("VISREP_HALO_TILE");
        v = (e && atoi(e) == 8) ? 8 : 16;                       // 16 x 8 tiles measured 3 % slower (profiles/round5_sd15_kernel_stats.md section 1)
        g_halo_tile.store(v, std::memory_order_relaxed);
    }
    return v;
}
}  // namespace

extern "C" int visrep_set_conv_halo_tile(int rows) {             // 16 (default: one workgroup per CU) | 8 (two); process-wide; returns the previous value
    if (rows != 8 && rows != 16) return visrep_set_error(VISREP_ERR_ARG, "conv_halo tile rows must be 8 or 16");
    const int old = halo_tile_rows();
    g_halo_tile.store(rows, std::memory_order_relaxed);
    return old;
}

extern "C" int visrep_conv3x3_halo_supported(int B, int H, int W, int C, int Cout) {
    return (B > 0 && C == HC && (Cout == 128 || Cout == 256) && H >= HT && W >= HT && H % HT == 0 && W % HT == 0 &&
            (long)B * H * W * HC < (1L << 40)) ? 1 : 0;
}

extern "C" int visrep_groupnorm_table_from_stats(const void* stats, const float* gamma, const float* beta, void* table, int B, int C, int groups, void* stream) {
    if (!stats || !gamma || !beta || !table) return visrep_set_error(VISREP_ERR_ARG, "groupnorm_table: null pointer");
    if (B <= 0 || C <= 0 || groups <= 0 || C % groups) return visrep_set_error(VISREP_ERR_SHAPE, "groupnorm_table: bad shape");
    const int BC = B * C;
    hipLaunchKernelGGL(gn_table_kernel, dim3((BC + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float2*)stats, gamma, beta, (float2*)table, BC, C, C / groups);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "groupnorm_table: launch failed");
}

extern "C" int visrep_conv3x3_bf16_halo(const void* x, int B, int H, int W, int C, const void* Wt, int ldw, const float* bias, void* out, int ldc, int Cout,
                                        int epilogue, const void* resid, const void* gn_table, int silu, void* gn_partial, int groups_out, void* stream) {
    if (!x || !Wt || !out) return visrep_set_error(VISREP_ERR_ARG, "conv3x3_halo: null pointer");
    if (!visrep_conv3x3_halo_supported(B, H, W, C, Cout))
        return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_halo: needs C = 128, Cout in {128, 256}, H and W multiples of 16 (visrep_conv3x3_halo_supported)");
    if (epilogue != VISREP_EPI_BIAS && epilogue != VISREP_EPI_RESID) return visrep_set_error(VISREP_ERR_ARG, "conv3x3_halo: epilogue must be BIAS or RESID");
    if (epilogue == VISREP_EPI_RESID && !resid) return visrep_set_error(VISREP_ERR_ARG, "conv3x3_halo: EPI_RESID needs resid");
    if (ldw < 9 * C || (ldw & 7) || (ldc & 7) || ldc < Cout) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_halo: ldw >= 9 C, ldw and ldc multiples of 8");
    if (((uintptr_t)x | (uintptr_t)Wt | (uintptr_t)out | (uintptr_t)resid | (uintptr_t)gn_table) & 15)
        return visrep_set_error(VISREP_ERR_ARG, "conv3x3_halo: pointers must be 16-byte aligned");
    int cpg = 0;
    if (gn_partial) {
        if (groups_out <= 0 || Cout % groups_out) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_halo: Cout must be a multiple of groups_out");
        cpg = Cout / groups_out;
        if (cpg != 4 && cpg != 8 && cpg != 16) return visrep_set_error(VISREP_ERR_SHAPE, "conv3x3_halo: GroupNorm partials need 4, 8 or 16 channels per group");
    }
    HaloArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)Wt; a.bias = bias; a.resid = (const bf16_t*)resid; a.out = (bf16_t*)out;
    a.gn_tab = (const float2*)gn_table; a.gn_partial = (float2*)gn_partial;
    a.B = B; a.H = H; a.W = W; a.ldw = ldw; a.ldc = ldc; a.silu = silu; a.gn_cpg = cpg;
    hipStream_t s = (hipStream_t)stream;
    visrep_count_route(VISREP_ROUTE_CONV_HALO);
    const bool rs = epilogue == VISREP_EPI_RESID;
    if (Cout == 128 && halo_tile_rows() == 8) return rs ? launch_halo<4, 2, true, 8>(a, s) : launch_halo<4, 2, false, 8>(a, s);
    if (Cout == 128) return rs ? launch_halo<4, 4, true>(a, s) : launch_halo<4, 4, false>(a, s);
    return rs ? launch_halo<8, 2, true>(a, s) : launch_halo<8, 2, false>(a, s);
}
