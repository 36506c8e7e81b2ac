// C score on gfx950: dense-correspondence keypoint transfer + PCK counting for zero-shot SPair-71k evaluation.
//
// Reference path (C_score/): normalize_feats (pck_train.py:24-29) -> sim = D1 D2^T (utils_correspondence.py:360)
// -> get_flow window soft-argmax (:297-337, :234-256) -> keypoint coordinates (:363-380) -> per-image PCK
// (pck_train.py:149-163).  The reference builds the full [P^2, P^2] Gram and gathers K <= 30 keypoint rows; only
// those K rows are computed here.
//
// cscore_transfer: one workgroup per image pair.
//   phase A  rows[k][t] = sum_c F1[c][s_k] * F2[c][t]  with v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain):
//            A operand = gathered source columns (k = keypoint), B operand = target map rows, read straight from
//            the [C, P^2] fp32 feature maps (128-B coalesced per half wave); squared norms of both operands are
//            accumulated from the same registers, so normalisation costs no extra HBM pass.
//            Each wave owns 32-target tiles; HBM traffic per pair = both maps once = 2 * P^2 * C * 4 B.
//            Layout 1 (position-major [P^2, C], the towers' own token layout): a keypoint's descriptor is one contiguous row, so
//            the source gather reads K * C * 4 B instead of K * C cache LINES (with [C, P^2] every (channel, keypoint) element
//            sits in its own 128-B line: 2.6 MB of line traffic per pair against 1 MB for the whole target map, and each wave
//            repeats it per tile); each lane streams 64 contiguous bytes of its row per step, a lane pair a full line.
//   phase B  per keypoint row (one wave each): first-index argmax, clamped (2w+1)^2 window, entries outside the
//            window are ZERO (not -inf) and stay in the softmax (reference semantics), beta = 0.02,
//            expectation over linspace(-1,1,P), un-normalise, clamp, scale to the annotation frame.
// cscore_pck: per pair hit counts at the three alpha thresholds; the comparison is done in fp64 exactly as the
//            reference's float32-alpha x float64-threshold promotion does.
#include "common.h"
#include "visrep_internal.h"

namespace {

struct CArgs {
    const float* feats;            // feature bank [n_images][C][P*P] fp32 (layout 0) or [n_images][P*P][C] (layout 1)
    const int* img1; const int* img2;   // per pair image index into the bank
    const int* patch_idx;          // [n_pairs][kmax] source patch index per keypoint
    const int* nkp;                // [n_pairs] keypoints in this pair (<= 32)
    const float* lin;              // [P] float32(np.linspace(-1, 1, P))
    float* xy;                     // [n_pairs][kmax][2] (x, y) in the annotation frame
    int n_pairs, kmax, P, C, split, window, soft;
    float beta, stride, half;      // anno_size / P, floor(stride / 2)
};

// phase B of the transfer, one wave per key-point similarity row [PP] (LDS): first-index argmax, clamped (2w+1)^2 window, entries outside
// the window are ZERO and stay in the softmax (reference semantics), expectation over linspace(-1, 1, P), annotation frame
VR_DEV void soft_argmax_row(const CArgs& p, const float* row, int PP, int lane, float* o) {
    const int w = p.window;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = lane; t < PP; t += 64) {
        const float v = row[t];
        if (v > bv) { bv = v; bi = t; }                        // per lane t increases -> first index kept on ties
    }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) {
        const float ov = __shfl_xor(bv, o2);
        const int oi = __shfl_xor(bi, o2);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    float ox, oy;
    if (p.soft) {
        const int mx = bi % p.P, my = bi / p.P;
        int x0 = 0, x1 = p.P - 1, y0 = 0, y1 = p.P - 1;
        if (w > 0) {
            x0 = max(mx - w, 0); x1 = min(mx + w, p.P - 1);
            y0 = max(my - w, 0); y1 = min(my + w, p.P - 1);
        }
        const bool has_out = (x1 - x0 + 1) * (y1 - y0 + 1) < PP;
        // w < 0: the "kernel soft-argmax" (apply_gaussian_kernel, utils_correspondence.py:278-295): a Gaussian of sigma = -w patches around the argmax
        // target weights every similarity.  The weight is 1 at the argmax and < 1 elsewhere, so the weighted maximum is bv when bv >= 0 and lies in
        // (bv, 0) otherwise: max(bv, 0) bounds it from above - the same stabiliser as with zeros outside a window.
        const float Mx = (has_out || w < 0) ? fmaxf(bv, 0.f) : bv;
        const float two_s2 = 2.f * (float)(w * w);
        float z = 0.f, ex = 0.f, ey = 0.f;
        for (int t = lane; t < PP; t += 64) {
            const int ty = t / p.P, tx = t - ty * p.P;
            const bool in = tx >= x0 && tx <= x1 && ty >= y0 && ty <= y1;
            float v = in ? row[t] : 0.f;
            if (w < 0) { const float dx = (float)(tx - mx), dy = (float)(ty - my); v *= expf(-(dx * dx + dy * dy) / two_s2); }
            const float e = expf((v - Mx) / p.beta);
            z += e; ex += e * p.lin[tx]; ey += e * p.lin[ty];
        }
        z = wave_sum(z); ex = wave_sum(ex); ey = wave_sum(ey);
        const float pm1 = (float)(p.P - 1);
        float fx = (ex / z + 1.f) * pm1 / 2.0f;
        float fy = (ey / z + 1.f) * pm1 / 2.0f;
        ox = fminf(fmaxf(fx, 0.f), pm1);
        oy = fminf(fmaxf(fy, 0.f), pm1);
    } else {
        ox = (float)(bi % p.P);
        oy = (float)(bi / p.P);
    }
    if (lane == 0) {
        o[0] = ox * p.stride + p.half;
        o[1] = oy * p.stride + p.half;
    }
}

template <bool PM>                 // PM: position-major bank [P^2, C]
__global__ __launch_bounds__(256) void cscore_transfer(const CArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int PP = p.P * p.P;
    float* rows = sm;                     // [32][PP]
    const int pair = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const int K = p.nkp[pair];
    const size_t map_elems = (size_t)p.C * PP;
    const float* F1 = p.feats + (size_t)p.img1[pair] * map_elems;
    const float* F2 = p.feats + (size_t)p.img2[pair] * map_elems;
    const int sk = (lq < K) ? p.patch_idx[(size_t)pair * p.kmax + lq] : 0;

    // ---------------- phase A
    // Channel range [0, split) / [split, C): two encoders concatenated on the channel axis are normalised SEPARATELY, then
    // the concatenation is normalised again (pck_train_two.py:24-36); split == 0 is the single-encoder path of pck_train.py.
    const int ntile = (PP + 31) >> 5;
    const bool two = p.split > 0;
    const int c_mid = two ? p.split : p.C;
    const float* a_ptr = PM ? F1 + (size_t)sk * p.C : F1 + (size_t)hi * PP + sk;
    float fa = 0.f, fb = 0.f;             // per keypoint (lane lq): source-side factors of the two channel blocks
    bool first = true;
    auto gram = [&](int c0, int c1, const float* bp, f32x16& acc, float& n1, float& n2) {
        int c = c0;
        if (PM) {
            // 32 channels per step: lane (lq, hi) streams channels [c + 16 hi, c + 16 hi + 16) of ITS row as four float4; the MFMA
            // k-pair of step (u, e) is (c + 4u + e, c + 16 + 4u + e) - any pairing is fine as long as both operands use it
            const float* ar = a_ptr + 16 * hi;
            const float* br = bp + 16 * hi;
            for (; c + 32 <= c1; c += 32) {
                float4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = *reinterpret_cast<const float4*>(ar + c + 4 * u);
                    b[u] = *reinterpret_cast<const float4*>(br + c + 4 * u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, bw[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        n1 += av[e] * av[e];
                        n2 += bw[e] * bw[e];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bw[e], acc, 0, 0, 0);
                    }
                }
            }
            for (; c < c1; c += 2) {                         // fewer than 32 channels left: one k-pair (c, c + 1) per MFMA
                const float a = a_ptr[c + hi];
                const float b = bp[c + hi];
                n1 += a * a;
                n2 += b * b;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            return;
        }
        for (; c + 16 <= c1; c += 16) {                  // 16 independent loads in flight per lane before the MFMA chain
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = a_ptr[(size_t)(c + 2 * u) * PP];
                b[u] = bp[(size_t)(c + 2 * u) * PP];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                n1 += a[u] * a[u];
                n2 += b[u] * b[u];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            }
        }
        for (; c < c1; c += 2) {
            const float a = a_ptr[(size_t)c * PP];
            const float b = bp[(size_t)c * PP];
            n1 += a * a;
            n2 += b * b;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    };
    for (int tile = wave; tile < ntile; tile += 4) {
        const int t = tile * 32 + lq;
        const int tc = t < PP ? t : PP - 1;
        const float* b_ptr = PM ? F2 + (size_t)tc * p.C : F2 + (size_t)hi * PP + tc;
        f32x16 acc_a = f32x16{}, acc_b = f32x16{};
        float n1a = 0.f, n1b = 0.f, n2a = 0.f, n2b = 0.f;
        gram(0, c_mid, b_ptr, acc_a, n1a, n2a);
        if (two) gram(c_mid, p.C, b_ptr, acc_b, n1b, n2b);
        if (first) {                                      // every tile re-reads the same source columns: keep the first sums
            n1a += __shfl_xor(n1a, 32); n1b += __shfl_xor(n1b, 32);
            const float ia = 1.0f / (sqrtf(n1a) + 1e-10f);            // 1 / (|d1_a[s_k]| + eps)
            if (two) {
                const float ib = 1.0f / (sqrtf(n1b) + 1e-10f);
                const float inv = 1.0f / (sqrtf(n1a * ia * ia + n1b * ib * ib) + 1e-10f);
                fa = ia * inv; fb = ib * inv;
            } else {
                fa = ia;
            }
            first = false;
        }
        n2a += __shfl_xor(n2a, 32); n2b += __shfl_xor(n2b, 32);
        float ga = 1.0f / (sqrtf(n2a) + 1e-10f), gb = 0.f;
        if (two) {
            gb = 1.0f / (sqrtf(n2b) + 1e-10f);
            const float inv = 1.0f / (sqrtf(n2a * ga * ga + n2b * gb * gb) + 1e-10f);
            ga *= inv; gb *= inv;
        }
        // lane holds G[k = row(r,hi)][t]; the keypoint-side factors live in lane k of this wave
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * hi;
            float v = acc_a[r] * ga * __shfl(fa, k);
            if (two) v += acc_b[r] * gb * __shfl(fb, k);
            if (t < PP) rows[k * PP + t] = v;
        }
    }
    __syncthreads();

    // ---------------- phase B: one wave per keypoint row
    for (int k = wave; k < K; k += 4) soft_argmax_row(p, rows + k * PP, PP, lane, p.xy + ((size_t)pair * p.kmax + k) * 2);
}

// ---- packed variant (position-major banks): one workgroup = one 32-row MFMA tile filled with the key points of SEVERAL pairs that share
// a target image.  A pair alone fills 11.5 of the 32 rows on average (K ~ U{3..20}) and every pair streams the whole target map: packing
// the pairs of a target (host side, cscore_ops.pack_rows) runs ~2.3x fewer tiles and target-map passes; consecutive groups share a target
// and are mapped onto ONE XCD (xcd_remap), so the second / third group of a target finds the map in that XCD's L2 (round 3 measured 7
// fabric fetches per map: the seven pairs of a target landed on seven XCDs).
// rows_tab[g][r] = (pair, k, source image, source patch index), pair < 0 = empty row; tgt[g] = target image.
struct CPackArgs {
    CArgs c;
    const int4* rows_tab;
    const int* tgt;
    int n_groups;
};

__global__ __launch_bounds__(256) void cscore_transfer_packed(const CPackArgs q) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const CArgs& p = q.c;
    const int PP = p.P * p.P;
    float* rows = sm;                     // [32][PP]
    const int group = xcd_remap(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const size_t map_elems = (size_t)p.C * PP;
    const int4 ent = q.rows_tab[(size_t)group * 32 + lq];                    // this lane's key-point row
    const float* F2 = p.feats + (size_t)q.tgt[group] * map_elems;
    // empty rows read the target's first descriptor (finite data, results never used)
    const float* a_ptr = ent.x >= 0 ? p.feats + (size_t)ent.z * map_elems + (size_t)ent.w * p.C : F2;

    const int ntile = (PP + 31) >> 5;
    const bool two = p.split > 0;
    const int c_mid = two ? p.split : p.C;
    float fa = 0.f, fb = 0.f;
    bool first = true;
    auto gram = [&](int c0, int c1, const float* bp, f32x16& acc, float& n1, float& n2) {
        int c = c0;
        const float* ar = a_ptr + 16 * hi;
        const float* br = bp + 16 * hi;
        for (; c + 32 <= c1; c += 32) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = *reinterpret_cast<const float4*>(ar + c + 4 * u);
                b[u] = *reinterpret_cast<const float4*>(br + c + 4 * u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, bw[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    n1 += av[e] * av[e];
                    n2 += bw[e] * bw[e];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bw[e], acc, 0, 0, 0);
                }
            }
        }
        for (; c < c1; c += 2) {
            const float a = a_ptr[c + hi];
            const float b = bp[c + hi];
            n1 += a * a;
            n2 += b * b;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    };
    for (int tile = wave; tile < ntile; tile += 4) {
        const int t = tile * 32 + lq;
        const int tc = t < PP ? t : PP - 1;
        const float* b_ptr = F2 + (size_t)tc * p.C;
        f32x16 acc_a = f32x16{}, acc_b = f32x16{};
        float n1a = 0.f, n1b = 0.f, n2a = 0.f, n2b = 0.f;
        gram(0, c_mid, b_ptr, acc_a, n1a, n2a);
        if (two) gram(c_mid, p.C, b_ptr, acc_b, n1b, n2b);
        if (first) {
            n1a += __shfl_xor(n1a, 32); n1b += __shfl_xor(n1b, 32);
            const float ia = 1.0f / (sqrtf(n1a) + 1e-10f);
            if (two) {
                const float ib = 1.0f / (sqrtf(n1b) + 1e-10f);
                const float inv = 1.0f / (sqrtf(n1a * ia * ia + n1b * ib * ib) + 1e-10f);
                fa = ia * inv; fb = ib * inv;
            } else {
                fa = ia;
            }
            first = false;
        }
        n2a += __shfl_xor(n2a, 32); n2b += __shfl_xor(n2b, 32);
        float ga = 1.0f / (sqrtf(n2a) + 1e-10f), gb = 0.f;
        if (two) {
            gb = 1.0f / (sqrtf(n2b) + 1e-10f);
            const float inv = 1.0f / (sqrtf(n2a * ga * ga + n2b * gb * gb) + 1e-10f);
            ga *= inv; gb *= inv;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * hi;
            float v = acc_a[r] * ga * __shfl(fa, k);
            if (two) v += acc_b[r] * gb * __shfl(fb, k);
            if (t < PP) rows[k * PP + t] = v;
        }
    }
    __syncthreads();
    for (int k = wave; k < 32; k += 4) {
        const int4 e = q.rows_tab[(size_t)group * 32 + k];                    // uniform
        if (e.x < 0) continue;
        soft_argmax_row(p, rows + k * PP, PP, lane, p.xy + ((size_t)e.x * p.kmax + e.y) * 2);
    }
}

// per pair: vis = v1*v2 > 0 ; err = |gt - pred|_2 (fp32) ; hit[a] = err < alpha[a](fp32->fp64) * thr(fp64)
__global__ void cscore_pck(const float* __restrict__ xy, const float* __restrict__ kps1, const float* __restrict__ kps2,
                           const double* __restrict__ thr, const int* __restrict__ nkp, int n_pairs, int kmax, float a0,
                           float a1, float a2, int* __restrict__ counts) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= n_pairs) return;
    const int K = nkp[pair];
    const double th = thr[pair];
    const double t0 = (double)a0 * th, t1 = (double)a1 * th, t2 = (double)a2 * th;
    int c0 = 0, c1 = 0, c2 = 0, nv = 0;
    for (int k = 0; k < K; ++k) {
        const float* k1 = kps1 + ((size_t)pair * kmax + k) * 3;
        const float* k2 = kps2 + ((size_t)pair * kmax + k) * 3;
        if (k1[2] * k2[2] > 0.f) {
            const float* pr = xy + ((size_t)pair * kmax + k) * 2;
            const float dx = k2[0] - pr[0], dy = k2[1] - pr[1];
            const float err = sqrtf(dx * dx + dy * dy);
            ++nv;
            c0 += (double)err < t0; c1 += (double)err < t1; c2 += (double)err < t2;
        }
    }
    int* o = counts + (size_t)pair * 4;
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = nv;
}

// ---- ADAPT_FLIP support (pck_train.py:111-126): the "mutual nearest neighbour" distance of two descriptor sets,
// utils_correspondence.py:54-73 get_distance_mutual_nn: D = cdist(F1, F2) over the L2-normalised descriptors; nn12[i] = argmin_j D[i, j],
// nn21[j] = argmin_i D[i, j]; result = mean over the mutual pairs (nn21[nn12[i]] == i) of D[i, nn12[i]].
// Input: the RAW Gram G = F1raw F2raw^T of the pair (visrep_gram_pairs_f32) and the normalisation factors r = 1 / (|x| + 1e-10) of both
// sets (eps = the 1e-10 of normalize_feats); with a = |x| r = 1 - eps r the squared distance is a_i^2 + b_j^2 - 2 G_ij r1_i r2_j (torch.cdist's matmul form, clamped at 0).
// One workgroup per pair, one wave per row (lanes over the columns, coalesced); the column minima are accumulated on the fly:
// lane l owns columns l + 64 k and keeps their running (min, first argmin) over the rows its wave visits in increasing order.
// Ties resolve to the smallest index like torch.argmin.
template <int KC>                          // columns per lane: PP <= 64 * KC (16: maps up to 32 x 32; 40: 48 x 48 diffusion maps; 64: 60 x 60)
__global__ __launch_bounds__(256) void mutual_nn_kernel(const float* __restrict__ gram, const float* __restrict__ r1, const float* __restrict__ r2,
                                                        int PP, float eps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dmin = sm;                          // [PP] min distance^2 of row i
    int* nn12 = reinterpret_cast<int*>(sm + PP);          // [PP]
    float* cmin = sm + 2 * PP;                 // [4][PP] per-wave column minima
    int* cidx = reinterpret_cast<int*>(sm + 6 * PP);      // [4][PP]
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* G = gram + (size_t)pair * PP * PP;
    const float* ra = r1 + (size_t)pair * PP;
    const float* rb = r2 + (size_t)pair * PP;
    float cb[KC], cv[KC];
    int ci[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int j = lane + 64 * k;
        cb[k] = j < PP ? rb[j] : 0.f;
        cv[k] = INFINITY; ci[k] = 0x7fffffff;
    }
    for (int i = wave; i < PP; i += 4) {
        const float ri = ra[i];
        const float a = 1.0f - eps * ri, a2 = a * a;
        float best = INFINITY;
        int bj = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int j = lane + 64 * k;
            if (j < PP) {
                const float b = 1.0f - eps * cb[k];
                const float d2 = fmaxf(a2 + b * b - 2.0f * G[(size_t)i * PP + j] * ri * cb[k], 0.f);
                if (d2 < best) { best = d2; bj = j; }              // increasing j within a lane: first minimum kept
                if (d2 < cv[k]) { cv[k] = d2; ci[k] = i; }           // increasing i within a wave: first minimum kept
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {                         // lexicographic (value, index) minimum over the lanes
            const float ov = __shfl_xor(best, o);
            const int oj = __shfl_xor(bj, o);
            if (ov < best || (ov == best && oj < bj)) { best = ov; bj = oj; }
        }
        if (lane == 0) { dmin[i] = best; nn12[i] = bj; }
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int j = lane + 64 * k;
        if (j < PP) { cmin[wave * PP + j] = cv[k]; cidx[wave * PP + j] = ci[k]; }
    }
    __syncthreads();
    int* nn21 = cidx;                                               // wave 0's slice receives the combined result
    for (int j = threadIdx.x; j < PP; j += 256) {
        float v = cmin[j];
        int idx = cidx[j];
        for (int w = 1; w < 4; ++w) {
            const float ov = cmin[w * PP + j];
            const int oi = cidx[w * PP + j];
            if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        nn21[j] = idx;
    }
    __syncthreads();
    float sum = 0.f;
    int cnt = 0;
    for (int i = threadIdx.x; i < PP; i += 256)
        if (nn21[nn12[i]] == i) { sum += sqrtf(dmin[i]); ++cnt; }
    sum = wave_sum(sum);
    float fc = wave_sum((float)cnt);
    __syncthreads();
    if (lane == 0) { cmin[wave] = sum; cmin[4 + wave] = fc; }
    __syncthreads();
    if (threadIdx.x == 0) out[pair] = (cmin[0] + cmin[1] + cmin[2] + cmin[3]) / (cmin[4] + cmin[5] + cmin[6] + cmin[7]);   // 0 / 0 = nan like torch's empty mean
}

// r[row] = 1 / (|x_row| + eps): normalize_feats' factor (pck_train.py:24-29), one wave per row
__global__ __launch_bounds__(256) void row_rnorm_kernel(const float* __restrict__ x, long rows, int C, float eps, float* __restrict__ r) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if (lane == 0) r[row] = 1.0f / (sqrtf(s) + eps);
}

}  // namespace

extern "C" int visrep_row_rnorm_f32(const float* x, long rows, int C, float eps, float* r, void* stream) {
    if (!x || !r) return visrep_set_error(VISREP_ERR_ARG, "row_rnorm_f32: null pointer");
    if (rows <= 0) return 0;
    if (C <= 0 || (C & 3)) return visrep_set_error(VISREP_ERR_SHAPE, "row_rnorm_f32: C must be a multiple of 4");
    hipLaunchKernelGGL(row_rnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, C, eps, r);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "row_rnorm_f32: launch failed");
}

template <int KC>
static void launch_mutual_nn(const float* gram, const float* r1, const float* r2, int n_pairs, int PP, float eps, float* out, hipStream_t st) {
    const size_t lds = (size_t)10 * PP * sizeof(float);
    static VisrepLdsOptIn opt;                                   // per (kernel instantiation, device): the largest size opted in so far
    if (lds > 48 * 1024) visrep_lds_opt_in(opt, reinterpret_cast<const void*>(mutual_nn_kernel<KC>), (int)lds);
    hipLaunchKernelGGL(mutual_nn_kernel<KC>, dim3(n_pairs), dim3(256), lds, st, gram, r1, r2, PP, eps, out);
}

extern "C" int visrep_mutual_nn_distance(const float* gram, const float* r1, const float* r2, int n_pairs, int PP, float eps, float* out, void* stream) {
    if (!gram || !r1 || !r2 || !out) return visrep_set_error(VISREP_ERR_ARG, "mutual_nn_distance: null pointer");
    if (n_pairs <= 0) return 0;
    if (PP <= 0 || PP > 4096) return visrep_set_error(VISREP_ERR_SHAPE, "mutual_nn_distance: 1 <= P*P <= 4096");
    hipStream_t st = (hipStream_t)stream;
    if (PP <= 1024) launch_mutual_nn<16>(gram, r1, r2, n_pairs, PP, eps, out, st);
    else if (PP <= 2560) launch_mutual_nn<40>(gram, r1, r2, n_pairs, PP, eps, out, st);
    else launch_mutual_nn<64>(gram, r1, r2, n_pairs, PP, eps, out, st);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "mutual_nn_distance: launch failed");
}

// ---- nearest-neighbour search of the MASK-based flip distance (utils_correspondence.py:41-50 get_distance: for every source cell inside its mask
// the Euclidean distance to the nearest target cell).  Operands arrive as the reference builds them - resized, masked, zeros replaced by
// -100000 - so distances are computed difference FIRST (line 47: norm(tgt - src)): a Gram-matrix form would cancel 1e10-sized squares.  A
// workgroup owns a 64 x 64 tile of (source cell, target cell) pairs, 4 x 4 per thread, channels staged through LDS sixteen at a time; the
// running minimum of a source row is kept as the bit pattern of a non-negative float (order-preserving), one atomicMin per row and tile.
__global__ __launch_bounds__(256) void masked_nn_min(const float* __restrict__ S, const float* __restrict__ T, unsigned* __restrict__ out, int ns, int nt, int C) {
    __shared__ float sS[16][64], sT[16][64];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int s0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const int lr = tid >> 2, lc = (tid & 3) * 4;                   // this thread stages row lr, channels lc .. lc + 3 of both tiles
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 16) {
        float4 vs = float4{0.f, 0.f, 0.f, 0.f}, vt = float4{0.f, 0.f, 0.f, 0.f};      // rows / channels past the end: zeros on both sides (difference 0)
        if (c0 + lc < C) {                                             // C % 4 == 0
            if (s0 + lr < ns) vs = *reinterpret_cast<const float4*>(S + (size_t)(s0 + lr) * C + c0 + lc);
            if (t0 + lr < nt) vt = *reinterpret_cast<const float4*>(T + (size_t)(t0 + lr) * C + c0 + lc);
        }
        sS[lc][lr] = vs.x; sS[lc + 1][lr] = vs.y; sS[lc + 2][lr] = vs.z; sS[lc + 3][lr] = vs.w;
        sT[lc][lr] = vt.x; sT[lc + 1][lr] = vt.y; sT[lc + 2][lr] = vt.z; sT[lc + 3][lr] = vt.w;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(&sS[k][ty * 4]), b = *reinterpret_cast<const float4*>(&sT[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = bv[j] - av[i]; acc[i][j] = __builtin_fmaf(d, d, acc[i][j]); }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m = 3.0e38f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (t0 + tx * 4 + j < nt) m = fminf(m, acc[i][j]);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) m = fminf(m, __shfl_xor(m, o));       // the sixteen threads of a source-row group are consecutive lanes
        if (tx == 0 && s0 + ty * 4 + i < ns) atomicMin(out + s0 + ty * 4 + i, __float_as_uint(m));
    }
}

extern "C" int visrep_masked_nn_min_f32(const float* src, const float* tgt, int ns, int nt, int C, float* min_d2, void* stream) {
    if (ns <= 0) return 0;
    if (!src || !tgt || !min_d2) return visrep_set_error(VISREP_ERR_ARG, "masked_nn_min: null pointer");
    if (nt <= 0 || C <= 0 || (C & 3)) return visrep_set_error(VISREP_ERR_SHAPE, "masked_nn_min: need targets and C % 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(min_d2, 0x7f, (size_t)ns * sizeof(float), st) != hipSuccess)      // 0x7f7f7f7f = 3.39e38: above every squared distance of finite operands
        return visrep_set_error(VISREP_ERR_LAUNCH, "masked_nn_min: memset failed");
    hipLaunchKernelGGL(masked_nn_min, dim3((nt + 63) / 64, (ns + 63) / 64), dim3(256), 0, st, src, tgt, reinterpret_cast<unsigned*>(min_d2), ns, nt, C);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "masked_nn_min: launch failed");
}

extern "C" int visrep_cscore_transfer(const float* feats, const int* img1, const int* img2, const int* patch_idx, const int* nkp,
                                      const float* lin, float* xy, int n_pairs, int kmax, int P, int C, int split, int window,
                                      int soft_eval, float beta, float anno_stride, float anno_half, int layout, void* stream) {
    if (n_pairs <= 0) return 0;
    if (layout != 0 && layout != 1) return visrep_set_error(VISREP_ERR_ARG, "cscore: layout must be 0 ([C, P*P]) or 1 ([P*P, C])");
    if (layout == 1 && ((C & 3) || (split & 3))) return visrep_set_error(VISREP_ERR_SHAPE, "cscore: the position-major layout needs C and split to be multiples of 4");
    if (kmax <= 0 || kmax > 32) return visrep_set_error(VISREP_ERR_SHAPE, "cscore: kmax must be in 1..32");
    if (P <= 0 || P > 32 || C <= 0 || (C & 1)) return visrep_set_error(VISREP_ERR_SHAPE, "cscore: need 1 <= P <= 32 and even C");
    if (split < 0 || split >= C || (split & 1)) return visrep_set_error(VISREP_ERR_SHAPE, "cscore: split must be even and in [0, C)");
    CArgs a{feats, img1, img2, patch_idx, nkp, lin, xy, n_pairs, kmax, P, C, split, window, soft_eval, beta, anno_stride, anno_half};
    const size_t lds = sizeof(float) * ((size_t)32 * P * P);
    static VisrepLdsOptIn opt[2];                                // per (layout, device)
    if (layout) visrep_lds_opt_in(opt[1], reinterpret_cast<const void*>(cscore_transfer<true>), (int)lds);
    else visrep_lds_opt_in(opt[0], reinterpret_cast<const void*>(cscore_transfer<false>), (int)lds);
    if (layout) hipLaunchKernelGGL(cscore_transfer<true>, dim3(n_pairs), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(cscore_transfer<false>, dim3(n_pairs), dim3(256), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "cscore_transfer: launch failed");
}

extern "C" int visrep_cscore_transfer_packed(const float* feats, const int* rows_tab, const int* tgt, const float* lin, float* xy, int n_groups, int kmax,
                                             int P, int C, int split, int window, int soft_eval, float beta, float anno_stride, float anno_half,
                                             void* stream) {
    if (n_groups <= 0) return 0;
    if (!feats || !rows_tab || !tgt || !lin || !xy) return visrep_set_error(VISREP_ERR_ARG, "cscore_packed: null pointer");
    if ((C & 3) || (split & 3)) return visrep_set_error(VISREP_ERR_SHAPE, "cscore_packed: position-major banks need C and split to be multiples of 4");
    if (kmax <= 0 || kmax > 32) return visrep_set_error(VISREP_ERR_SHAPE, "cscore: kmax must be in 1..32");
    if (P <= 0 || P > 32 || C <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "cscore: need 1 <= P <= 32");
    if (split < 0 || split >= C) return visrep_set_error(VISREP_ERR_SHAPE, "cscore: split must be in [0, C)");
    if ((uintptr_t)rows_tab & 15) return visrep_set_error(VISREP_ERR_ARG, "cscore_packed: rows_tab must be 16-byte aligned");
    CPackArgs a{{feats, nullptr, nullptr, nullptr, nullptr, lin, xy, 0, kmax, P, C, split, window, soft_eval, beta, anno_stride, anno_half},
                reinterpret_cast<const int4*>(rows_tab), tgt, n_groups};
    const size_t lds = sizeof(float) * ((size_t)32 * P * P);
    static VisrepLdsOptIn opt;
    visrep_lds_opt_in(opt, reinterpret_cast<const void*>(cscore_transfer_packed), (int)lds);
    hipLaunchKernelGGL(cscore_transfer_packed, dim3(n_groups), dim3(256), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "cscore_transfer_packed: launch failed");
}

extern "C" int visrep_pck_count(const float* xy, const float* kps1, const float* kps2, const double* thresholds, const int* nkp,
                                int n_pairs, int kmax, const float* alphas3, int* counts, void* stream) {
    if (n_pairs <= 0) return 0;
    hipLaunchKernelGGL(cscore_pck, dim3((n_pairs + 127) / 128), dim3(128), 0, (hipStream_t)stream, xy, kps1, kps2, thresholds, nkp,
                       n_pairs, kmax, alphas3[0], alphas3[1], alphas3[2], counts);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "pck_count: launch failed");
}
