/* libvisrep_hip.so — C ABI of the MI355X (gfx950) vision-representation scoring path.
 *
 * Drop-in boundary: the reference is pure Python on PyTorch; its hot path is a chain of stock
 * PyTorch calls.  Each entry point below replaces one such call chain and is what a ctypes binding inside the
 * reference would load.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor .data_ptr()) unless marked HOST
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is enqueued
 *     asynchronously on it; nothing here synchronises, allocates or frees device memory
 *   - return value 0 = ok, negative = error (VISREP_ERR_*); visrep_last_error() gives the message; no C++
 *     exception crosses the boundary; no process-global mutable state: the error string and the A/B variant knobs are thread-local,
 *     caller-owned scratch is keyed by (device, stream), per-kernel LDS opt-ins and CU counts are cached per device
 *   - bf16 tensors are raw uint16 storage (torch.bfloat16), row-major, leading dimensions in ELEMENTS
 */
#ifndef VISREP_H
#define VISREP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VISREP_VERSION 600   /* round 6: reserved CUs, GEMM tile-walk knob, gemm variants 6-8, visrep_masked_nn_min_f32, window < 0 in the C-score transfers (new exports / newly accepted argument values only).  500 = round 5: device CU count, routing counters, MFMA probe, A-score reference arithmetic, host twins (new exports only).  410 = round 4: split-route `products`, stream-keyed scratch, per-thread knobs (changed signatures); 410: q_prescaled = 2, row-mapped GEMM, image-aligned / wide-head attention, conv + GroupNorm partials */

enum { VISREP_BF16 = 0, VISREP_F32 = 1 };
enum { VISREP_OK = 0, VISREP_ERR_ARG = -1, VISREP_ERR_SHAPE = -2, VISREP_ERR_LAUNCH = -3 };
/* GEMM epilogues */
enum {
    VISREP_EPI_BIAS = 0,   /* C = A W^T + bias                                   (nn.Linear)                    */
    VISREP_EPI_ACT = 1,    /* C = act(A W^T + bias)                              (Linear + QuickGELU/GELU)      */
    VISREP_EPI_RESID = 2,  /* C = resid + ls * (A W^T + bias)                    (Linear + LayerScale + add)    */
    VISREP_EPI_VT = 3,     /* V^T scatter for visrep_mhsa_fwd (see below)                                        */
    VISREP_EPI_PATCH = 4,  /* patch-embed: row remap past the CLS row + position add                             */
    VISREP_EPI_F32 = 5     /* C(fp32) = A W^T + bias                                                             */
};
enum { VISREP_ACT_NONE = 0, VISREP_ACT_QUICK_GELU = 1, VISREP_ACT_GELU_ERF = 2, VISREP_ACT_GELU_TANH = 3 };

int visrep_version(void);
/* copies the calling thread's last error message (NUL terminated) into buf; returns its length */
size_t visrep_last_error(char* buf, size_t n);

/* Kernel-variant knobs for A/B measurements.  They are PER-THREAD state (the calling thread's later launches only): there is no
 * process-global mutable state behind this ABI, and the defaults need no call at all.  Each returns the previous value.
 *
 * GEMM family used by every entry point below: 1 = 128x128 tiles, one barrier per K-tile (also: N % 256 != 0 shapes, tail rows, split-K,
 * implicit 3x3 convolution); 2 = 256x256 persistent ping-pong kernel, 64-byte LDS rows (runs when K % 64 != 0); 5 (default) = the same
 * structure with 128-byte LDS rows, a five-slot LDS-DMA ring and half the barriers (profiles/round2_gemm_v5.md).  2 and 5 need
 * N % 256 == 0.  Results are identical up to fp32 summation order.  (Variants 3 and 4 - measured dead ends, profiles/round2_gemm_v4.md -
 * exist only in the tools-only VISREP_EXPERIMENTS build.)  6 / 7 / 8 (round 6, A/B only: measured 20-30 % slower, profiles/round6_gemm.md) =
 * the "duo" kernel, two independent 4-wave workgroups per CU on 256 x 128 tiles (N % 128 == 0, K % 64 == 0, K >= 128; BIAS / ACT / RESID / VT
 * epilogues, everything else runs as variant 5): 6 = wherever it applies, 7 = for N <= 1024 only, 8 = its unpipelined first build.  Bitwise
 * equal to variant 5. */
int visrep_set_gemm_variant(int variant);
/* Attention forward at head width 64: 1 (default) = attn_fwd, the four-wave kernel that also serves head widths 128 / 192.  2 = attn_fwd_ab
 * (two 32-row query blocks per wave; measured equal-to-slower, profiles/round3_attention.md) exists only in the VISREP_EXPERIMENTS build;
 * the production library refuses it. */
int visrep_set_attn_variant(int variant);
/* Tile shape of the bf16 A-score Gram (visrep_ascore_maxcos*): 0 (default) = whichever launches the smaller tile area for (Nt, Nr);
 * 1 = one 128 x 128 tile per workgroup; 2 = persistent ping-pong tiles of 192 or 256 rows per operand (the GEMM default's structure;
 * 576 = 3 x 192, 256 = 1 x 256).  Results are identical up to fp32 summation order (none: both sum k in the same order per MFMA chain). */
int visrep_set_ascore_variant(int variant);
/* Timing-only ablation of GEMM variant 2 (bit 0: skip MFMAs, bit 1: skip the LDS-DMA loads, bit 2: skip the fragment reads):
 * results are WRONG for mask != 0; used by tools/gemm_ablate.py to attribute cycles.  Returns the previous mask. */
int visrep_debug_gemm_ablation(int mask);
/* Diagnostic builds (-DVISREP_GEMM_ABLATE) only: device buffer of 16 uint64 receiving per-segment cycle sums of variant 2
 * (wave 0 and wave 4 of block 0); ignored by production builds. */
int visrep_debug_gemm_timing_buffer(void* dev_u64x16);
/* Routing counters of the CALLING THREAD: how many launches each kernel family received since the last reset (host-side integers,
 * incremented by the dispatchers; HIP-graph replays do not pass through a dispatcher and are not counted).  Tests use them to assert
 * that a shape takes the route it was tuned for - e.g. that the 768-px diffusion tower's 256- / 512-channel convolutions run in the
 * persistent 256x256 kernel and its 128-channel ones in the 128x128 kernel with GroupNorm partials.  out: VISREP_ROUTE_COUNT longs
 * (NULL: only reset); reset != 0 clears the counters after the copy.  Returns VISREP_ROUTE_COUNT. */
enum {
    VISREP_ROUTE_GEMM_256 = 0,       /* persistent 256x256 kernel (variants 2 / 5), plain GEMM: head launch or whole problem */
    VISREP_ROUTE_GEMM_128 = 1,       /* 128x128 kernel, plain GEMM (N % 256 != 0, few tiles, variant 1) */
    VISREP_ROUTE_GEMM_TAIL = 2,      /* rows of a partial last tile round split off to the 128x128 kernel */
    VISREP_ROUTE_SPLITK = 3,         /* deterministic split-K pair (partials + reduce), GEMM or convolution */
    VISREP_ROUTE_CONV_256 = 4,       /* implicit 3x3 convolution in the persistent 256x256 kernel */
    VISREP_ROUTE_CONV_128 = 5,       /* implicit 3x3 convolution in the 128x128 kernel */
    VISREP_ROUTE_CONV_128_GN = 6,    /* ... that also emits GroupNorm partial sums */
    VISREP_ROUTE_ATTN = 7,           /* attn_fwd<ND> (head width 64 / 128 / 192) */
    VISREP_ROUTE_ATTN_WIDE = 8,      /* attn_fwd_wide (head width 512) */
    VISREP_ROUTE_ATTN_CLS = 9,       /* attn_fwd_cls (image-aligned CLS towers) */
    VISREP_ROUTE_CONV_HALO = 10,     /* conv3x3_halo: halo-resident 3x3 convolution with the input's GroupNorm + SiLU fused */
    VISREP_ROUTE_CONV_C8 = 11,       /* conv3x3_c8: the VAE's first convolution straight from 8-channel pixel tokens (no im2col) */
    VISREP_ROUTE_COUNT = 12
};
int visrep_debug_routes(long* out, int reset);
/* Matrix-pipe ceiling of THIS device under THIS process' conditions: `iters` bursts of 32 dependent-free v_mfma_f32_16x16x32_bf16 per wave,
 * 8 waves per CU, every CU, operands in registers (pseudo-random bf16 with |x| in [0.25, 4) when random != 0, small constants otherwise), no memory
 * traffic.  FLOP per launch = CUs * 8 * iters * 32 * 16384 (returned through *flop when non-NULL); the caller times the launch on
 * `stream`.  bench.py quotes the result beside the 2.5 PFLOP/s nominal peak: on random data the chip's power management holds a pure
 * MFMA stream near 1.9-2.0 PFLOP/s (DVFS), which is the practical roof every fraction on the line can also be read against.
 * sink: >= 24 bytes of 8-byte-aligned device memory: bytes [0, 4) are never written in practice; u64 [1] = shader-clock cycles (s_memtime) and
 * u64 [2] = 100-MHz ticks (s_memrealtime) of block 0's MFMA loop, i.e. the clock the stream ran at = [1] / ([2] * 10 ns) - the same pair of
 * counters, read in the tile-timing build of the GEMM (tools/gemm_tile_timing.py), gives the clock the GEMM runs at (profiles/round5_gemm.md). */
int visrep_debug_mfma_probe(int iters, int random, void* sink, double* flop, void* stream);
/* XCD-weighted tile split of the persistent 256x256 GEMM / convolution kernel (round 5).  The eight XCDs of an MI355X run at their own clocks
 * under the power limit (+-4 % around the mean inside one launch, measured with s_memtime / s_memrealtime: profiles/round5_gemm.md), so equal
 * shares of the tile list end when the slowest XCD does - and which XCD that is also depends on the kernel and its shape.  Per (kernel
 * instantiation, M, N, K) one launch in 8 records when each XCD finished; its last block leaves the time per round of tiles of each XCD in
 * pinned host memory; the following launches of that shape give every XCD whole rounds in proportion to its speed.  Results are bitwise
 * independent of the split.  OFF by default: a round of tiles is 2.8 % of a block's work at the headline shapes, as large as the effect, and
 * the same-box A/B measured +0.1 .. +0.3 % on the forward (profiles/round5_gemm.md section 3).  visrep_set_xcd_balance(1) (or
 * VISREP_XCD_BALANCE=1 in the environment) switches it on, process-wide; returns the previous setting.  Launches that are being CAPTURED into a
 * HIP graph always run with equal shares and are never measurements (a plan baked into a graph would be replayed with capture-time bounds forever):
 * the split serves eager launches only - the ViT / diffusion engines' graph replays are unaffected by it.  visrep_debug_xcd_balance: the smoothed relative time per round of each XCD (1 = mean; rel8 may
 * be NULL) of the shape launched most recently on the current device, and the number of measurements folded in so far; returns the setting. */
int visrep_set_xcd_balance(int on);
int visrep_debug_xcd_balance(float* rel8, unsigned* updates);
/* The split itself (pure host arithmetic, no device needed): rel8 = time per round of tiles of each XCD (any positive scale), grid = persistent
 * blocks (a multiple of 8), ntiles >= grid -> bounds9[x] .. bounds9[x + 1] = the tile indices of XCD x, bounds9[8] = ntiles; whole rounds of
 * grid / 8 tiles in proportion to speed, leftovers to the XCD that would finish first.  Returns 0, -1 for unusable arguments. */
int visrep_debug_xcd_split(const float* rel8, int grid, int ntiles, int* bounds9);

/* ---- optional device scratch owned by the caller (e.g. one torch tensor kept alive for the process).  With it,
 * visrep_gemm_bf16 splits the K loop of problems that have few output tiles but a deep reduction (the diffusion towers'
 * 3x3 convolutions at 12x12 / 24x24 resolution) across CUs and reduces the fp32 partial planes in slice order, i.e.
 * deterministically.  Registrations are keyed by (device, stream): visrep_set_scratch registers the buffer every stream of the CURRENT
 * device falls back to (enough when one stream per device runs GEMMs at a time); visrep_set_stream_scratch registers a buffer for ONE
 * stream of the current device and wins over the device-wide one - a caller that runs split-K GEMMs on several streams of a device
 * concurrently gives each stream its own (up to 8 per device).  (NULL, 0) detaches.  Both are thread-safe. */
int visrep_set_scratch(void* ptr, size_t bytes);
int visrep_set_stream_scratch(void* stream, void* ptr, size_t bytes);
/* Compute units of the CURRENT device as the dispatchers count them (cached per device; 256 on MI355X).  Callers that size launches for
 * whole tile rounds (engine.best_chunk, the sweep's launch plan) use it so that their arithmetic is the dispatcher's on any part. */
int visrep_device_cu_count(void);
/* CUs the persistent kernels leave free (round 6; multi-GPU insurance).  With world > 1 the sweep's C leg runs RCCL's all_to_all WHILE the next
 * tower launch's persistent GEMMs (grid = CU count) would own every CU (sweep.c_score_of; the reference has no counterpart: its feature dump,
 * llava/feature/extract.py:198-214, has no collective at all).  k (rounded down to a multiple of 8: the same number per XCD; ignored if fewer than
 * 64 CUs would remain) is subtracted from what visrep_device_cu_count() and every dispatcher see; 0 = off (default).  Process-wide;
 * VISREP_RESERVE_CUS=k in the environment is the same switch.  Set it before engines capture their HIP graphs.  Returns the previous value.
 * Unmeasured: no multi-GPU box was available in rounds 1-6; it exists so that the first one can A/B it. */
int visrep_set_reserved_cus(int k);
/* Tile order of the persistent 256x256 GEMM kernel (round 6 A/B knob, per thread; results do not depend on it).  0 = default: XCD-contiguous
 * chunks of the tile list, row-major, 4 x 8 windows for more than 8 column panels.  C = 1..255 (low byte): column-group-major - every XCD owns
 * whole row panels and walks them once per group of C column panels, so that the group's weight panels stay in that XCD's 4-MB L2 across rounds
 * (taken when the tile list splits into whole row panels per XCD and N / 256 is a multiple of C, else the default order).  Measured in
 * profiles/round6_gemm.md (FETCH_SIZE and time per window shape); the default stays 0.  Returns the previous value. */
int visrep_set_gemm_walk(int code);

/* ---- dense layers: replaces torch.nn.functional.linear (+ bias / activation / residual) inside
 * HF CLIPEncoderLayer / Dinov2Layer / SiglipEncoderLayer (transformers, called from
 * llava/model/multimodal_encoder/clip_encoder.py:48) and the mm_projector Sequential
 * (llava/model/multimodal_projector/builder.py:40-47).
 * A [M,K] bf16 (lda), W [N,K] bf16 (ldw, nn.Linear layout), bias [N] fp32 or NULL, C [M,N] bf16 (fp32 for EPI_F32).
 * N % 128 == 0, K % 64 == 0.  resid (bf16, same shape/ld as C, may alias C) and ls (fp32 [N] or NULL) are used by
 * EPI_RESID only. */
int visrep_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K,
                     int epilogue, int act, const void* resid, const float* ls, void* stream);

/* ---- LayerNorm over the last dim (torch.nn.LayerNorm in the HF blocks; CLIP pre_layrnorm).  x,y bf16 [rows,d];
 * gamma/beta fp32 [d]; fp32 statistics; y may alias x. */
int visrep_layernorm(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int rows, int d, float eps,
                     void* stream);

/* ---- LayerNorm folded into a GEMM.  visrep_layernorm_stats only READS x and writes rt[row] = (rstd, -mean * rstd) (float2 per row;
 * allocate round_up(rows, 128) + 8 entries, zero past `rows`); visrep_gemm_bf16_ln then computes, for A = the RAW rows x and
 * W = gamma o W_orig,   C[m, n] = epilogue( rt[m].x * (A W^T)[m, n] + rt[m].y * ln_s[n] + bias[n] )
 * which equals Linear(LayerNorm(x)) when ln_s[n] = sum_k W[n, k] and bias = W_orig beta + b.  epilogue: VISREP_EPI_BIAS, _ACT or _VT.
 * The normalised activations are never written to HBM (the reference materialises them: HF nn.LayerNorm before every
 * attention / MLP block, e.g. modeling_clip.py CLIPEncoderLayer.forward). */
int visrep_layernorm_stats(const void* x, int ldx, void* rt, int rows, int d, float eps, void* stream);
int visrep_gemm_bf16_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const void* ln_rt, const float* ln_s, void* C,
                        int ldc, int M, int N, int K, int epilogue, int act, void* stream);
/* Residual GEMM (VISREP_EPI_RESID of visrep_gemm_bf16) that ALSO leaves the LayerNorm statistics of its output rows in
 * rt[m] = (rstd, -mean * rstd) over the N columns — what the next block's visrep_gemm_bf16_ln consumes, so that LayerNorm never
 * reads the residual stream again (HF blocks: x = x + attn(ln1(x)); x = x + mlp(ln2(x))).  The 256x256 kernel emits per-row partial
 * sums of the bf16-rounded outputs from its epilogue (partial: scratch of M * N / 64 float2) which are reduced in a fixed order;
 * shapes routed elsewhere (split-K, 128x128 tail rows) run visrep_layernorm_stats on the rows they wrote.  N <= 2048. */
/* A GEMM over a periodic subset of A's rows: logical row r reads physical row (r / row_period) * row_stride + r % row_period + row_first
 * (e.g. the patch tokens of every image: period T - 1, stride T, first 1 - the reference's feature_select `[:, 1:]`,
 * clip_encoder.py:37-44 - or the CLS rows: period 1, stride T, first 0) without gathering them first.  ln_rt / ln_s (both or neither):
 * folded LayerNorm, statistics indexed by PHYSICAL row.  Epilogues BIAS, ACT, VT (row_period % 4 == 0), F32. */
int visrep_gemm_bf16_rows(const void* A, int lda, int row_period, int row_stride, int row_first, const void* W, int ldw, const float* bias,
                          const void* ln_rt, const float* ln_s, void* C, int ldc, int M, int N, int K, int epilogue, int act, void* stream);
int visrep_gemm_bf16_resid_stats(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K,
                                 const void* resid, const float* ls, void* rt, void* partial, float eps, void* stream);

/* ---- multi-head self-attention forward (HF CLIPAttention / Dinov2SelfAttention / SiglipAttention: softmax(QK^T
 * scale) V, fp32 softmax).  qk: [B*T, 2*H*64] bf16 (Q | K, head-major columns); vt: V^T as written by
 * visrep_gemm_bf16(..., VISREP_EPI_VT): [H*64, ldvt] with ldvt >= round_up(B*T, 64), % 64 == 0, columns beyond B*T
 * finite; out: [B*T, H*64] bf16.  head_dim must be 64.  scale <= 0: Q already carries scale * log2(e) (folded into the Q projection by
 * the caller): the kernel computes p = 2^(q.k - reference) with the reference subtracted inside the matrix pipe (accumulator init). */
int visrep_mhsa_fwd(const void* qk, int ldqk, const void* vt, int ldvt, void* out, int ldo, int B, int T, int H, int head_dim,
                    float scale, void* stream);

/* ---- image-aligned self-attention of a CLS tower whose patch count is a multiple of 64 (T = 1 + 64 n: CLIP-L/14-336 577, the 224-px
 * L/14 towers 257), pre-scaled Q only.  qk as for visrep_mhsa_fwd; vt: V^T of the PATCH tokens only, [H*64, ldvt], column
 * b (T - 1) + t - 1 (perm16 inside 16-token blocks, as VISREP_EPI_VT writes it), ldvt >= B (T - 1); vcls: the V rows of the CLS tokens
 * [B, ldvc] row-major.  An image's keys are then whole 64-key tiles (no masks, one tile less per image than the global tiling of
 * visrep_mhsa_fwd) and the CLS key enters as the initial state of the online softmax (m = q.k_cls, l = 1, O = v_cls).
 * visrep_mhsa_cls_supported(T) says whether T qualifies.  Same semantics as visrep_mhsa_fwd(scale <= 0). */
int visrep_mhsa_cls_supported(int T);
int visrep_mhsa_cls_fwd(const void* qk, int ldqk, const void* vt, int ldvt, const void* vcls, int ldvc, void* out, int ldo, int B, int T,
                        int H, int head_dim, void* stream);

/* ---- general multi-head attention forward for the diffusion towers' transformer blocks (vendored diffusers
 * attention_processor.py AttnProcessor2_0 / SlicedAttnProcessor called from attention.py:241-283 BasicTransformerBlock):
 * self-attention (k/vt built from the same tokens, Tk = Tq) and cross-attention to the prompt (Tk = text length).
 * q: [B*Tq, ldq] bf16, head h in columns [h*head_dim, (h+1)*head_dim); k: [B*Tk, ldk] likewise ([Tk, ldk] when
 * kv_shared = 1: one key/value sequence - the prompt - serves every batch item); vt: [H*head_dim, ldvt] as written by
 * VISREP_EPI_VT over the key rows, ldvt >= round_up(key rows, 64); out: [B*Tq, ldo].  head_dim in {64, 128, 192}: the
 * weight packer zero-pads narrower heads (SD1.5: 40 / 80 / 160), which changes nothing in softmax(QK^T)V.  head_dim 512 (the single head of
 * the diffusion VAE's mid-block attention, AutoencoderKL: 9216 latent pixels at 768 px) runs a wide-head kernel that needs whole key tiles
 * (Tk % 64 == 0), causal = 0 and scale > 0.
 * causal = 1 masks keys after the query position (HF CLIPTextTransformer's causal mask: the prompt encoder behind
 * pipe.encode_prompt, dift_sd.py:258-263).  scale <= 0 (head_dim 64 only): pre-scaled Q, as for visrep_mhsa_fwd. */
int visrep_attention_fwd(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* out, int ldo, int B, int Tq,
                         int Tk, int H, int head_dim, int kv_shared, int causal, float scale, void* stream);

/* ---- patch embedding pieces (HF CLIPVisionEmbeddings / Dinov2Embeddings / SiglipVisionEmbeddings) */
int visrep_im2col(const void* pixels, int pixel_dtype, void* cols, int B, int Himg, int Wimg, int patch, int Kpad, void* stream);
int visrep_cls_rows(void* x, int ldx, const float* cls, const float* pos, int B, int T, int d, void* stream);
int visrep_cast_f32_bf16(const float* src, void* dst, long n, void* stream);

/* ---- composed tower forward: replaces `self.vision_tower(images, output_hidden_states=True).hidden_states[k]`
 * (clip_encoder.py:48-49, dinov2_encoder.py:51-52, siglip_encoder.py:49-50). */
typedef struct {
    int image_size, patch, d, heads, mlp, layers, tokens, has_cls, pre_ln, act, kpad;
    float eps;
    /* 1: the Q rows of wqkv (and of bqkv) carry head_dim^-0.5 * log2(e) - folded in by the weight packer (the bf16 engine does it while it
     * folds LayerNorm: same single rounding of the fp32 product) - and the attention kernel exponentiates the raw scores (visrep_mhsa_fwd
     * with scale <= 0).  2: as 1, and towers with a CLS token and 64 n patches use the image-aligned kernel (visrep_mhsa_cls_fwd: V
     * projected by two row-mapped GEMMs, patch rows -> V^T, CLS rows -> [B, d]); other shapes behave as 1.  bf16 tower only; the fp32
     * towers ignore it. */
    int q_prescaled;
} visrep_vit_config;
typedef struct {           /* device pointers; matrices bf16 [out,in], vectors fp32; ls1/ls2 NULL when no LayerScale */
    const float *ln1_g, *ln1_b; const void* wqkv; const float* bqkv; const void* wo; const float* bo; const float* ls1;
    const float *ln2_g, *ln2_b; const void* w1; const float* b1; const void* w2; const float* b2; const float* ls2;
    /* LayerNorm folded into the consuming GEMMs (both non-NULL to enable; ln*_g / ln*_b are then unused): wqkv = gamma1 o Wqkv and
     * bqkv = Wqkv beta1 + b (likewise w1 / b1 with LN2), sqkv[n] = sum_k wqkv[n, k] over the bf16-rounded rows [3d], s1 [mlp]. */
    const float *sqkv, *s1;
} visrep_vit_layer;
typedef struct {
    const void* patch_w;   /* bf16 [d, kpad], zero padded past 3*p*p */
    const float *patch_b, *cls, *pos, *pre_ln_g, *pre_ln_b;   /* patch_b / cls / pre_ln_* may be NULL */
    const visrep_vit_layer* layers;   /* HOST array of `layers` entries */
} visrep_vit_weights;
size_t visrep_vit_workspace_bytes(const visrep_vit_config* cfg, int B);
/* hidden: [B, tokens, d] bf16 — receives hidden_states[n_layers] (the residual stream lives in it). */
int visrep_vit_forward(const visrep_vit_config* cfg, const visrep_vit_weights* w, const void* pixels, int pixel_dtype, void* hidden,
                       int B, int n_layers, void* workspace, void* stream);

/* ---- reference-precision (fp32) tower path.  The reference builds the C-score CLIP / OpenCLIP / DINOv2 towers with no dtype cast and
 * fp32 pixels (C_score/extract_feature.py:36-45,49-50,80-87): every tensor and accumulation here is fp32 (v_mfma_f32_32x32x2_f32 = an
 * exact fmaf chain, 157 TFLOP/s peak), so images -> tower -> A / C scores can be held to the 1e-4 bar against the fp32 reference
 * chain.  A parity mode, ~1/20 of the bf16 engine's rate.
 * visrep_gemm_f32: C = epi(alpha * A W^T + bias); A [M,K] (lda), W [N,K] (w_kn = 0, nn.Linear) or [K,N] (w_kn = 1), all fp32, rows
 * 16-byte aligned (pointers, leading dimensions, batch strides % 4 floats); epilogue BIAS / ACT / RESID (C = resid + ls * (..), resid
 * may alias C).  Batched over nb1 x nb2 problems with element strides strides6 = {A1, A2, W1, W2, C1, C2} (NULL when nb1 = nb2 = 1):
 * replaces torch.matmul(q, k.transpose) / torch.matmul(attn, v) of HF's eager attention as well as F.linear. */
int visrep_gemm_f32(const float* A, int lda, const float* W, int ldw, int w_kn, const float* bias, float* C, int ldc, int M, int N, int K,
                    int epilogue, int act, const float* resid, const float* ls, float alpha, int nb1, int nb2, const long* strides6,
                    void* stream);
/* nn.LayerNorm in fp32 (two-pass variance); y may alias x */
int visrep_layernorm_f32(const float* x, int ldx, const float* g, const float* b, float* y, int ldy, int rows, int d, float eps, void* stream);
/* in-place softmax over the first `cols` entries of every row (ld >= cols) */
int visrep_softmax_rows_f32(float* x, int ld, long rows, int cols, void* stream);
/* composed fp32 forward = visrep_vit_forward with fp32 pixels, fp32 MATRICES in the weight struct (sqkv / s1 ignored) and an fp32
 * hidden [B, tokens, d]; any head width that is a multiple of 4. */
size_t visrep_vit_f32_workspace_bytes(const visrep_vit_config* cfg, int B);
int visrep_vit_forward_f32(const visrep_vit_config* cfg, const visrep_vit_weights* w, const float* pixels, float* hidden, int B, int n_layers,
                           void* workspace, void* stream);
/* ---- fp32 on the bf16 matrix pipe ("split-bf16"): an fp32 value x is carried as bf16 planes hi = bf16(x), mid = bf16(x - hi),
 * lo = bf16(x - hi - mid) (the subtractions are exact; two planes hold 16 significand bits, three hold 24) and a product sum runs over
 * `products` plane pairs, accumulated in fp32 by v_mfma_f32_16x16x32_bf16:
 *   6 = (hi,hi) (hi,mid) (hi,lo) (mid,hi) (mid,mid) (lo,hi): every term >= 2^-24 of the result; differs from the exact fp32 result by ~1e-7
 *       relative - below the rounding of an fp32 FMA chain - at 16/6 of the exact-fp32 MFMA rate;
 *   4 = the full product of two-plane operands; 3 = (hi,hi) (hi,mid) (mid,hi): terms below 2^-16 dropped, ~4e-6 relative per product sum
 *       (between TF32 and fp32), at 16/3 of the exact-fp32 rate; accuracy against A / PCK in profiles/round4_precision.md.
 * The reference-precision towers (C_score/extract_feature.py:36-45: CLIP / OpenCLIP / DINOv2 in fp32) run their projections this way;
 * tests/test_gpu_f32.py holds the route to the same bars as the exact-fp32 one. */
/* planes [rows, nplanes K] bf16 = hi | mid [| lo] of x fp32 [rows, K] (ldx floats per row); nplanes 2 | 3; K, ldx % 4 == 0 */
int visrep_split_bf16_planes(const float* x, int ldx, long rows, int K, int nplanes, void* planes, void* stream);
/* C [M, N] fp32 (ldc; may be NULL) and / or out_planes [M, npl N] (may be NULL) = epilogue(A W^T): v = act(A W^T + bias);
 * v = resid + ls * v when resid != NULL (resid fp32 [M, ldc], may alias C; ls [N] or NULL = 1).  products 3 | 4 | 6; npl = 3 for 6 products,
 * else 2; a_planes [M, npl K], w_planes [N, npl K] from visrep_split_bf16_planes; N % 256 == 0, K % 64 == 0; act = VISREP_ACT_* evaluated
 * with libm expf / erff / tanhf. */
int visrep_gemm_f32_split(const void* a_planes, const void* w_planes, int M, int N, int K, int products, const float* bias, int act,
                          const float* resid, const float* ls, float* C, int ldc, void* out_planes, void* stream);
/* 1 when visrep_vit_forward_f32_split takes this tower (d, mlp % 256 == 0, head width 64) */
int visrep_vit_f32_split_supported(const visrep_vit_config* cfg);
/* visrep_vit_forward_f32 with the projections and the attention as split-bf16 products.  w: the fp32 weights (vectors, patch matrix); wsplit:
 * same struct whose wqkv / wo / w1 / w2 point to the planes [N, npl K] of the fp32 matrices (other fields ignored); products as above.
 * Same workspace size. */
int visrep_vit_forward_f32_split(const visrep_vit_config* cfg, const visrep_vit_weights* w, const visrep_vit_weights* wsplit, int products,
                                 const float* pixels, float* hidden, int B, int n_layers, void* workspace, void* stream);
/* Per-thread diagnostic of the fp32 towers' attention (tests only): bit 0 = three-launch attention (batched Q K^T -> softmax rows -> P V
 * through HBM) on the exact route; bit 1 = exact-fp32 MFMA attention inside the split route.  Returns the previous mask. */
int visrep_debug_f32_attention(int mask);

/* ---- fp32 convolution-block primitives of the supervised C-score post-processor (C_score/model_utils/projection_network.py:15-125
 * AggregationNetwork = detectron2-style BottleneckBlocks, model_utils/resnet.py:174-286: 1x1 / 3x3 / 1x1 bias-free convolutions, each
 * followed by nn.GroupNorm, ReLU, projection shortcut).  Tokens are channels-last fp32 [B, H*W, C]; a 1x1 convolution is visrep_gemm_f32,
 * a 3x3 one visrep_im2col3x3_f32 (pad 1, stride 1; column order (ky, kx, c)) + visrep_gemm_f32 on weights repacked [Cout, 9 C];
 * visrep_groupnorm_f32: y = alpha * act(GN(x) * gamma + beta + resid) (+ y when accumulate), resid may be NULL, relu 0 / 1. */
int visrep_im2col3x3_f32(const float* x, float* cols, int B, int H, int W, int C, void* stream);
int visrep_groupnorm_f32(const float* x, const float* gamma, const float* beta, const float* resid, float* y, int B, int HW, int C, int groups,
                         float eps, int relu, float alpha, int accumulate, void* stream);

/* ---- ADAPT_FLIP support of the C score (C_score/pck_train.py:111-126 with MUTUAL_NN; utils/utils_correspondence.py:54-73
 * get_distance_mutual_nn): raw Gram matrices of image pairs from a position-major fp32 bank [n_images, PP, C] (gather by index),
 * normalize_feats' row factors r = 1 / (|x| + eps) (pck_train.py:24-29), and per pair the mean cdist of the mutual nearest neighbours
 * of the two L2-normalised descriptor sets: out[z] fp32 (nan when a pair has none, like torch's empty mean).  gram [n_pairs, PP, PP],
 * r1 / r2 [n_pairs, PP] (the factors of each pair's source / target map, eps = the epsilon they were built with), PP <= 4096
 * (60 x 60 maps). */
int visrep_gram_pairs_f32(const float* bank, const int* idx1, const int* idx2, int n_pairs, int PP, int C, float* gram, void* stream);
int visrep_row_rnorm_f32(const float* x, long rows, int C, float eps, float* r, void* stream);
int visrep_mutual_nn_distance(const float* gram, const float* r1, const float* r2, int n_pairs, int PP, float eps, float* out, void* stream);
/* ADAPT_FLIP WITHOUT MUTUAL_NN (C_score/pck_train.py:122-124 -> utils/utils_correspondence.py:22-52 get_distance, the mask-based flip
 * distance; round 6): the search of lines 41-50 - for every source cell (row of src [ns, C]: the cells inside the source mask) the squared
 * Euclidean distance to the nearest of the nt target cells (rows of tgt [nt, C]), difference first as the reference computes it (its operands
 * hold -100000 where a mask or an exact zero was), -> min_d2 [ns] fp32.  The resize / mask / substitution steps of lines 23-37 are the host
 * side's (cscore_ops.masked_nn_distance); the mean of sqrt(min_d2) is the reference's return value.  C % 4 == 0. */
int visrep_masked_nn_min_f32(const float* src, const float* tgt, int ns, int nt, int C, float* min_d2, void* stream);

/* ---- JPEG decode for the device input pipeline.  Replaces PIL's Image.open(path).convert('RGB') of the reference's
 * image loaders (C_score/extract_feature.py:65-66, llava/mm_utils.py:78-95, llava/feature/extract.py:198-214) bit for bit; what PIL runs
 * underneath is libjpeg-turbo's default decode (jidctint.c islow IDCT, jdsample.c fancy upsampling, jdcolor.c ycc_rgb_convert).
 * HOST: visrep_jpeg_info parses the headers (rc != 0: not a decodable JPEG stream; rc == 0 and info->unsupported != 0: a valid file this
 * decoder does not take - progressive, arithmetic, 12-bit, CMYK, multi-scan, unusual chroma layouts - visrep_last_error() says which; the
 * Python side lets PIL decode such a file).  visrep_jpeg_entropy_decode: baseline Huffman decode into QUANTISED coefficients, int16,
 * natural 8x8 order, component planes of whole MCUs one after the other (info->coef_count values), and the components' quantisation
 * tables qtab[ncomp][64] (natural order).  Both run on the calling host thread, thread-safe, no GPU needed.
 * DEVICE: visrep_jpeg_reconstruct turns a BATCH of decoded images into packed RGB u8 [H, W, 3] each: dequantisation + islow IDCT into
 * the `planes` scratch, then fancy chroma upsampling + YCbCr -> RGB.  desc: n_images x 32 int64 (device) - coefficient / plane offsets
 * per component, block grid, component and image sizes, offsets of the image's RGB output and quantisation tables
 * (law_of_vision_representation_in_mllms_amd/device_jpeg.py builds it); max_blocks / max_pixels: the largest image's. */
typedef struct VisrepJpegInfo {
    int width, height, ncomp;          /* ncomp 1 (grey) or 3 (YCbCr) */
    int hs[3], vs[3], hmax, vmax;      /* sampling factors */
    int mcus_w, mcus_h;
    int blocks_w[3], blocks_h[3];      /* block grid per component (whole MCUs) */
    int comp_w[3], comp_h[3];          /* real samples per component (jdmaster.c downsampled_width / height) */
    int restart_interval, progressive, unsupported;
    long coef_count;
} VisrepJpegInfo;
int visrep_jpeg_info(const void* data, size_t n, VisrepJpegInfo* info);
int visrep_jpeg_entropy_decode(const void* data, size_t n, int16_t* coef, uint16_t* qtab);
int visrep_jpeg_reconstruct(const void* coef, const void* qtab, const void* desc, int n_images, long max_blocks, long max_pixels,
                            void* planes, void* rgb, void* stream);

/* ---- A score (A_score/compute.py:12-15,54-72): scores[img] = mean_t max_s cos(other[img][t], ref[img][s]).
 * other [n_img,Nt,D], ref [n_img,Nr,D] contiguous, dtype VISREP_BF16 (D%16==0) or VISREP_F32 (D%8==0);
 * scores fp32 [n_img]; workspace of visrep_ascore_workspace_bytes(). */
size_t visrep_ascore_workspace_bytes(int n_img, int Nt, int Nr);
int visrep_ascore_maxcos(const void* other, const void* ref, int n_img, int Nt, int Nr, int D, int dtype, float* scores,
                         void* workspace, void* stream);
/* The per-row factor of the two normalisations (normalize_feat's 1/(|x|+1e-10) and F.cosine_similarity's eps clamp, compute.py:12-15,
 * 64-65), scale[rows] fp32 — and the score with factors computed earlier (NULL = compute here, as visrep_ascore_maxcos does).  The
 * reference re-normalises the clip336 / clip224 reference sets for every encoder and every encoder's tokens once per reference
 * (compute.py:54-56 inside the loops); computed once per tensor instead, the norm passes drop from 29 % of the A score's kernel time. */
int visrep_ascore_row_scale(const void* x, long rows, int D, int dtype, float* scale, void* stream);
int visrep_ascore_maxcos_scaled(const void* other, const void* ref, const float* other_scale, const float* ref_scale, int n_img, int Nt,
                                int Nr, int D, int dtype, float* scores, void* workspace, void* stream);

/* The A score in the REFERENCE'S OWN ARITHMETIC on bf16 tensors: A_score/compute.py:12-15,54-72 executed by torch on bf16 inputs rounds the
 * result of EVERY op to bf16 (norm, divide, the element-wise products inside F.cosine_similarity, their sum, the mean) - which is what the
 * published table holds (policy/ablations_t.csv: 1.0078125 for CLIP336 against itself).  other / ref: bf16 [n_img, Nt | Nr, D] contiguous,
 * any D; scores fp32 [n_img], each a bf16-representable value as `.item()` of the reference's bf16 scalar; workspace of
 * visrep_ascore_refarith_workspace_bytes().  VALU kernel (the rounded products are not a matrix product): a parity mode. */
size_t visrep_ascore_refarith_workspace_bytes(int n_img, int Nt, int Nr, int D);
int visrep_ascore_maxcos_refarith(const void* other, const void* ref, int n_img, int Nt, int Nr, int D, float* scores, void* workspace,
                                  void* stream);

/* ---- C score (C_score/utils/utils_correspondence.py:345-382 calculate_keypoint_transformation with get_flow,
 * C_score/pck_train.py:24-29 normalize_feats): feats = bank of [C, P*P] fp32 maps; per pair image indices, source
 * patch indices [n_pairs,kmax] (kmax <= 32), keypoint counts; lin = float32(np.linspace(-1,1,P)); xy [n_pairs,kmax,2].
 * split = 0: one encoder (pck_train.py); split = C1 > 0: channels [0,C1) and [C1,C) are two encoders normalised separately,
 * concatenated and normalised again (C_score/pck_train_two.py:24-36 normalize_feats).
 * layout = 0: maps are [C, P*P] (the reference's on-disk [1, C, P, P]); layout = 1: [P*P, C], the towers' own token layout - a
 * keypoint's descriptor is then one contiguous row (C and split multiples of 4).
 * window: > 0 the (2w+1)^2 window around the argmax (get_flow :301-320: entries outside are ZERO and stay in the softmax); 0 the plain
 * soft-argmax; < 0 (round 6) the Gaussian-kernel soft-argmax (:321-324 -> apply_gaussian_kernel :278-295, sigma = -window patches; the
 * reference hard-wires a 60 x 60 grid there, these kernels take their own P). */
int visrep_cscore_transfer(const float* feats, const int* img1, const int* img2, const int* patch_idx, const int* nkp,
                           const float* lin, float* xy, int n_pairs, int kmax, int P, int C, int split, int window, int soft_eval,
                           float beta, float anno_stride, float anno_half, int layout, void* stream);
/* The same transfer for position-major banks ([n_images, P*P, C]) with the key points of SEVERAL pairs packed into one 32-row MFMA tile:
 * a group = pairs that share a target image, at most 32 key points in total (the caller packs them: one pair alone fills 11.5 of 32 rows
 * on SPair-71k).  rows_tab int32 [n_groups, 32, 4] = (pair, key-point index, source image, source patch index) per tile row, pair < 0 for an
 * empty row, 16-byte aligned; tgt int32 [n_groups] = the group's target image; xy [n_pairs, kmax, 2] as above (rows of a group write their own
 * (pair, key point) slot).  Same arithmetic per row as visrep_cscore_transfer, bit for bit. */
int visrep_cscore_transfer_packed(const float* feats, const int* rows_tab, const int* tgt, const float* lin, float* xy, int n_groups, int kmax,
                                  int P, int C, int split, int window, int soft_eval, float beta, float anno_stride, float anno_half,
                                  void* stream);
/* per-pair PCK hit counts (C_score/pck_train.py:101,149-163): kps1/kps2 [n_pairs,kmax,3] (x,y,vis) fp32, thresholds
 * fp64 [n_pairs], alphas3 = HOST pointer to 3 floats; counts int32 [n_pairs,4] = hits@a0,a1,a2, n_visible. */
int visrep_pck_count(const float* xy, const float* kps1, const float* kps2, const double* thresholds, const int* nkp, int n_pairs,
                     int kmax, const float* alphas3, int* counts, void* stream);

/* ==== HOST twins ("`*_cpu` twins with identical signatures minus stream back config 1"; BASELINE configs[0]: the CLIP-L/14 tower
 * on 32 images, CPU fp32, plumbing without a GPU).  Plain C++ on host threads, fp32, HOST pointers, `threads` <= 0 = min(hardware threads, 32).
 * An independent implementation (csrc/host_twins.hip): it shares no code with the device kernels and never runs unless the caller asks for
 * device "cpu" explicitly - the device entry points above still fail without a GPU.  Same semantics as their device counterparts:
 *   visrep_vit_forward_cpu: fp32 weights in the visrep_vit_weights struct (matrices fp32 [out, in], patch_w [d, kpad]; sqkv / s1 ignored),
 *   pixels fp32 [B, 3, S, S], hidden fp32 [B, tokens, d] = hidden_states[n_layers];
 *   visrep_ascore_maxcos_cpu: fp32 [n_img, Nt | Nr, D] -> scores fp32 [n_img];
 *   visrep_cscore_transfer_cpu / visrep_pck_count_cpu: arguments of visrep_cscore_transfer / visrep_pck_count. */
int visrep_vit_forward_cpu(const visrep_vit_config* cfg, const visrep_vit_weights* w, const float* pixels, float* hidden, int B, int n_layers,
                           int threads);
int visrep_ascore_maxcos_cpu(const float* other, const float* ref, int n_img, int Nt, int Nr, int D, float* scores, int threads);
int visrep_cscore_transfer_cpu(const float* feats, const int* img1, const int* img2, const int* patch_idx, const int* nkp, const float* lin,
                               float* xy, int n_pairs, int kmax, int P, int C, int split, int window, int soft_eval, float beta,
                               float anno_stride, float anno_half, int layout, int threads);
int visrep_pck_count_cpu(const float* xy, const float* kps1, const float* kps2, const double* thresholds, const int* nkp, int n_pairs, int kmax,
                         const float* alphas3, int* counts);

/* ==== Convolutional-tower primitives (Stable-Diffusion feature towers) ==================================
 * All activations are channels-last token matrices: [B*H*W, C] bf16.  A 3x3 convolution = visrep_im2col3x3 +
 * visrep_gemm_bf16 with K = ld of the gathered matrix and weights repacked [Cout, (ky, kx, Cin)] (zero-padded to ld). */

/* torch.nn.GroupNorm (+ optional SiLU) as used by diffusers resnet.py:ResnetBlock2D.forward (norm1/norm2 + nonlinearity),
 * transformer_2d.py:141 and vae.py conv_norm_out.  x, y: [B*HW, C] bf16 (y may alias x); gamma/beta fp32 [C]; fp32
 * statistics over (HW, C/groups) per (image, group), summed in a fixed order (bit-reproducible run to run).
 * workspace: visrep_groupnorm_workspace_bytes(B, HW, groups). */
size_t visrep_groupnorm_workspace_bytes(int B, int HW, int groups);
int visrep_groupnorm(const void* x, const float* gamma, const float* beta, void* y, int B, int HW, int C, int groups, float eps,
                     int silu, void* workspace, void* stream);

/* 3x3 patch gather for nn.Conv2d(kernel 3): y[(b, oy, ox), (ky*3+kx)*C + c] = x[b, oy*stride+ky-pad, ox*stride+kx-pad, c]
 * (zero outside), columns [9C, ldy) zeroed.  pad_mode 0 = padding 1 on every side (resnet convs, UNet downsample
 * stride 2: downsampling.py Downsample2D padding=1); pad_mode 1 = F.pad(x, (0,1,0,1)) then no padding (VAE encoder
 * downsample, downsampling.py:130-133).  upsample = 1 reads the source through a nearest-neighbour 2x upsample
 * (upsampling.py Upsample2D: F.interpolate(scale_factor=2, mode="nearest") + conv) without materialising it. */
int visrep_im2col3x3(const void* x, void* y, int B, int H, int W, int C, int stride, int pad_mode, int upsample, int ldy, void* stream);

/* nn.Conv2d(kernel 3) as an IMPLICIT GEMM: the gather of visrep_im2col3x3 happens inside the GEMM's A-operand LDS-DMA
 * (per-lane source addresses, zero page for padded taps), so the 9x-expanded patch matrix never exists in HBM.
 * x: [B*H*W, C] bf16 channels-last, C % 64 == 0; Wt: [Cout (% 64 == 0), ldw >= 9*C] bf16 with K order (ky, kx, c);
 * out: [B*Ho*Wo, ldc] (fp32 for VISREP_EPI_F32); stride / pad_mode / upsample as visrep_im2col3x3; epilogue BIAS, RESID
 * (out = resid + conv + bias, resid [B*Ho*Wo, ldc]) or F32.  Few-tile problems use the split-K path when a scratch is set. */
int visrep_conv3x3_bf16(const void* x, int B, int H, int W, int C, const void* Wt, int ldw, const float* bias, void* out, int ldc, int Cout,
                        int stride, int pad_mode, int upsample, int epilogue, const void* resid, void* stream);

/* The same convolution (no upsampling, epilogue BIAS | RESID) that ALSO emits the GroupNorm statistics of its output - the norm1 / norm2 /
 * conv_norm_out that follows it in diffusers resnet.py:ResnetBlock2D / vae.py:Encoder - as partial sums from its epilogue:
 * gn_partial[((b * (Ho Wo / 64) + slot) * groups + g)] = (sum, sum of squares) of the fp32 outputs of image b, rows [64 slot, 64 slot + 64), group g
 * (visrep_conv_gn_partial_bytes(B, Ho Wo, groups) bytes, every entry written).  visrep_groupnorm_from_partials() then normalises without
 * reading the tensor for its statistics.  Supported when visrep_conv_gn_supported_epi(B, Ho Wo, Cout, groups, epilogue): Ho Wo % 128 == 0,
 * Cout % 64 == 0, Cout / groups in {4, 8, 16}, and a kernel that emits the sums without costing the convolution its route: the 128x128
 * kernel (both epilogues, from its epilogue's store loop) or - round 5 - the 256x256 ping-pong kernel for VISREP_EPI_BIAS (a packed-math pass
 * over the accumulators in front of its epilogue); a RESID convolution that the 256x256 kernel runs is answered 0 and keeps the separate
 * statistics pass.  visrep_conv_gn_supported() is the round-4 query without the epilogue (the answer that holds for both epilogues). */
int visrep_conv_gn_supported(int B, int HWo, int Cout, int groups);
int visrep_conv_gn_supported_epi(int B, int HWo, int Cout, int groups, int epilogue);
size_t visrep_conv_gn_partial_bytes(int B, int HWo, int groups);
int visrep_conv3x3_bf16_gn(const void* x, int B, int H, int W, int C, const void* Wt, int ldw, const float* bias, void* out, int ldc, int Cout,
                           int stride, int pad_mode, int epilogue, const void* resid, void* gn_partial, int groups, void* stream);
/* GroupNorm (+ SiLU) of x [B*HW, C] from such partial sums (HW % 64 == 0): finalize + apply.  workspace: >= B * groups * 8 bytes. */
int visrep_groupnorm_from_partials(const void* x, const float* gamma, const float* beta, void* y, int B, int HW, int C, int groups, float eps,
                                   int silu, const void* partial, void* workspace, void* stream);

/* ---- 3x3 convolution (stride 1, padding 1) with the INPUT's GroupNorm (+ SiLU) fused into its operand path, for the VAE encoder's
 * 128-channel layers (diffusers resnet.py ResnetBlock2D.forward: norm1 -> SiLU -> conv1 -> norm2 -> SiLU -> conv2 (+ shortcut); vae.py
 * Encoder at 768^2 / 384^2).  A workgroup owns a 16 x 16 output tile: its 18 x 18 x 128 input halo is fetched ONCE, normalised in registers
 * (gn_table[b, c] = (scale, shift) = (rstd gamma, beta - mean rstd gamma), then SiLU when silu != 0 - groupnorm_apply's arithmetic bit for
 * bit; NULL = use x as it is) and kept in LDS; the nine taps are shifted views of it, only W streams.  No normalised copy of the tensor
 * ever exists in HBM (the separate apply pass reads and re-writes 2 x 2.4 GB per layer at 768^2 x 16 images).  Results equal
 * visrep_conv3x3_bf16 on the normalised tensor bit for bit (same K order, same MFMA chains).
 * x [B H W, 128] bf16 channels-last; Wt [Cout, ldw >= 1152] with K order (ky, kx, c); out / resid [B H W, ldc]; epilogue BIAS | RESID.
 * gn_partial (may be NULL): GroupNorm partial sums of the OUTPUT for the norm that follows, layout of visrep_conv3x3_bf16_gn
 * (visrep_conv_gn_partial_bytes(B, H W, groups_out) bytes; slot = 64 pixels of a 16 x 4 patch; consumed by visrep_groupnorm_stats_from_partials
 * / visrep_groupnorm_from_partials).  Supported: C = 128, Cout in {128, 256}, H and W multiples of 16 (visrep_conv3x3_halo_supported). */
int visrep_conv3x3_halo_supported(int B, int H, int W, int C, int Cout);
/* Tile height of the Cout = 128 kernel: 16 (default; 16 x 16-pixel tiles, 8 waves, 147 KB of LDS: one workgroup per CU) or 8 (16 x 8 tiles, 4 waves,
 * 77 KB: two workgroups per CU, one's halo phase under the other's K loop - measured 3 % slower, kept as a tested variant).  Results are bitwise the
 * same.  Process-wide (also VISREP_HALO_TILE=8 in the environment); returns the previous value, < 0 for another argument. */
int visrep_set_conv_halo_tile(int rows);
int visrep_conv3x3_bf16_halo(const void* x, int B, int H, int W, int C, const void* Wt, int ldw, const float* bias, void* out, int ldc, int Cout,
                             int epilogue, const void* resid, const void* gn_table, int silu, void* gn_partial, int groups_out, void* stream);

/* ---- The VAE encoder's first convolution (diffusers vae.py Encoder.conv_in: Conv2d(3, 128, 3, padding = 1); dift_sd.py:172 vae.encode) without
 * im2col: x [B*H*W, 8] bf16 tokens (visrep_nchw_to_tokens with Cpad = 8), Wt [128, ldw >= 96] in K order (ky, kx, c8) with zero padding columns (the
 * packer's im2col layout), out [B*H*W, ldc].  The MFMA's pixel operand is read straight from the neighbour pixels' 16-byte tokens (one global load
 * per lane and k-step, zeros outside the image), nothing is staged; the epilogue is the GEMM kernels' (bias, 16-byte stores) and optionally leaves
 * the GroupNorm partial sums of the output (gn_partial as visrep_conv3x3_bf16_gn: visrep_conv_gn_partial_bytes(B, H W, groups) bytes; NULL = none).
 * Supported: Cout = 128, W % 16 == 0, H W % 64 == 0 (% 128 with partial sums).  Equals visrep_im2col3x3 + visrep_gemm_bf16 up to the order of
 * the fp32 sum inside an MFMA. */
int visrep_conv3x3_c8_supported(int B, int H, int W, int Cout);
int visrep_conv3x3_c8_bf16(const void* x, int B, int H, int W, const void* Wt, int ldw, const float* bias, void* out, int ldc, int Cout,
                           void* gn_partial, int groups, void* stream);
/* The statistics half of GroupNorm on its own: stats[b, g] = (mean, rstd) fp32 pairs [B, groups], from a read-only pass over x
 * (visrep_groupnorm_stats; workspace of visrep_groupnorm_workspace_bytes) or from a producing convolution's partial sums
 * (visrep_groupnorm_stats_from_partials, HW % 64 == 0) - and the per-(image, channel) (scale, shift) table the fused convolution
 * consumes: table [B, C] float2 (visrep_groupnorm_table_from_stats). */
int visrep_groupnorm_stats(const void* x, void* stats, int B, int HW, int C, int groups, float eps, void* workspace, void* stream);
int visrep_groupnorm_stats_from_partials(const void* partial, void* stats, int B, int HW, int C, int groups, float eps, void* stream);
int visrep_groupnorm_table_from_stats(const void* stats, const float* gamma, const float* beta, void* table, int B, int C, int groups, void* stream);

/* activations.py GEGLU: y[m, f] = x[m, f] * gelu_erf(x[m, F + f]); x [M, >= 2F] bf16, y [M, >= F] bf16. */
int visrep_geglu(const void* x, int ldx, void* y, int ldy, long M, int F, void* stream);

/* softmax(scale * scores) per row: fp32 scores [rows, lds] -> bf16 probabilities [rows, ldp], columns [n, ldp) zeroed.
 * Used by the single-head 512-wide VAE mid-block attention (attention_processor.py Attention, vae.py:UNetMidBlock2D). */
int visrep_softmax_rows(const float* scores, int lds, void* probs, int ldp, int rows, int n, float scale, void* stream);

/* [B, C, H, W] (dtype VISREP_BF16 | VISREP_F32) -> [B*H*W, Cpad] bf16, channels [C, Cpad) zero. */
int visrep_nchw_to_tokens(const void* x, int dtype, void* y, int B, int C, int H, int W, int Cpad, void* stream);

/* F.interpolate(x, size=(OH, OW), mode="bilinear") (align_corners=False, no antialias) over `planes` = B*C contiguous
 * [H, W] planes (dtype VISREP_BF16 | VISREP_F32) -> bf16 [planes, OH, OW].  The image-variation featurizer resizes the
 * input to 224x224 for its CLIP image encoder this way (dift_imsd.py:215-216). */
int visrep_resize_bilinear(const void* x, int dtype, void* y, int planes, int H, int W, int OH, int OW, void* stream);

/* dift_sd.py:172-176: latents = (mean + exp(0.5 * clamp(logvar, -30, 20)) * post_noise) * scaling_factor
 * (autoencoder_kl.py encode + vae.py DiagonalGaussianDistribution.sample), then the scheduler's add_noise as
 * y = coef_latent * latents + coef_noise * ddim_noise:  DDIMScheduler (scheduling_ddim.py:471-495) has
 * coef_latent = sqrt(alphas_cumprod[t]), coef_noise = sqrt(1 - alphas_cumprod[t]); the vendored
 * FlowMatchEulerDiscreteScheduler.add_noise used by the SD3 featurizer (scheduling_flow_match_euler_discrete.py:192-210,
 * dift_sd3.py:108-111) has coef_latent = t, coef_noise = 1 - t with the raw integer timestep.  moments: fp32 [B*HW, ldm] with
 * the mean in columns [0, Z) and the log-variance in [Z, 2Z); the two noise tensors are the reference's randn draws made
 * explicit, fp32 [B, Z, H, W]; y: [B*HW, Cpad] bf16, channels [Z, Cpad) zero. */
int visrep_sd_noisy_latents(const float* moments, int ldm, const float* post_noise, const float* ddim_noise, void* y, int B, int Z,
                            int HW, int Cpad, float scaling, float coef_latent, float coef_noise, void* stream);

/* dift_sd.py:275 ensemble mean: x [B, E, N] bf16 -> y [B, N] bf16 (fp32 accumulation). */
int visrep_mean_groups(const void* x, void* y, int B, int E, long N, void* stream);

/* ==== Device input pipeline: everything after JPEG decode ================================================
 * One axis of Pillow's 8-bit separable resampling (src/libImaging/Resample.c ImagingResampleHorizontal_8bpc /
 * Vertical_8bpc) - what `PIL.Image.resize` does inside the reference's pre-processing (C_score/extract_feature.py:65-66,
 * HF CLIPImageProcessor.resize, llava/mm_utils.py:64-95) - BIT-EXACT: out[line, xx, c] = clip8((2^21 + sum_x in[line,
 * xmin[xx] + x, c] * kk[xx][x]) >> 22) with bounds = (xmin, count) pairs and the 22-bit fixed-point coefficient table the
 * host builds like precompute_coeffs + normalize_coeffs_8bpc (device_preprocess.pil_coeffs).  Strides in bytes select the
 * axis: horizontal pass line = row, vertical pass line = column. */
int visrep_resample_u8(const void* in, void* out, long n_lines, int out_len, int channels, long in_line_stride, long in_elem_stride,
                       long out_line_stride, long out_elem_stride, const int* bounds, const int* kk, int ksize, void* stream);

/* crop + ToTensor + Normalize: out[c, y, x] = (in[y0+y, x0+x, c] / 255 - mean3[c]) / std3[c], IEEE fp32 (bit-identical to the
 * numpy expression of the CPU processors); in: uint8 [H, W, 3] on the device; mean3 / std3: HOST float[3];
 * out: [3, crop_h, crop_w] VISREP_F32 or VISREP_BF16. */
int visrep_u8hwc_to_chw_norm(const void* in, int H, int W, int x0, int y0, int crop_h, int crop_w, const float* mean3, const float* std3,
                             void* out, int dtype, void* stream);

/* Both resampling passes + crop + ToTensor + Normalize for a BATCH of images of different sizes in two launches - what
 * `processor.preprocess(expand2square(Image.open(p).convert('RGB')))` (llava/feature/extract.py:198-214, llava/mm_utils.py:78-95) and
 * `Image.open(p).convert('RGB').resize((s, s))` + `(x / 255 - 0.5) * 2` (C_score/extract_feature.py:65-70) do per image, bit for bit.
 * desc: DEVICE int64 [n_images][24]: 0 source pointer (uint8 [H, W, 3] on the device), 1 H, 2 W, 3-4 canvas height / width (= H, W, or
 * the padded square of expand2square), 5-6 row / column of the image inside the canvas, 7 background colour r | g << 8 | b << 16,
 * 8 mirror the image left-right (pck_train.py:112), 9 pointer to this image's scratch [rows, crop_w, 3] for the horizontal pass,
 * 10-11 first canvas row / row count of that scratch (the rows the vertical pass reads), 12-14 horizontal bounds pointer / coefficient
 * pointer / ksize (ksize 0: no horizontal resize), 15-17 the same for the vertical pass, 18-19 crop origin x0, y0 in the resized image.
 * max_mid_rows = max over images of field 11 with ksize != 0 (0: no image needs a horizontal pass).  out: [n_images, 3, crop_h, crop_w]. */
int visrep_preprocess_u8_batch(const void* desc, int n_images, long max_mid_rows, int crop_h, int crop_w, const float* mean3,
                               const float* std3, void* out, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
