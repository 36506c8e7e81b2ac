// HOST twins of the composed entry points ("CPU twins with identical signatures minus stream (`*_cpu`) back config 1";
// BASELINE configs[0]: CLIP-L/14 tower on CPU fp32, "plumbing, no GPU").
//
// Plain C++ on host threads (std::thread, no OpenMP runtime dependency), fp32 throughout, HOST pointers.  An INDEPENDENT implementation:
// nothing here calls or shares code with oracle/ (test infrastructure) or with the device kernels; it exists so that the reference's
// plug-in surface can be exercised on a box without a GPU - selected ONLY by an explicit device="cpu" on the Python side
// (engine.VitEngineCPU, ascore_ops.max_cos_mean_cpu, cscore_ops.transfer_cpu), never as a fallback: every device entry point still fails
// loudly without a GPU (tests/test_abi_symbols.py).  Not a performance path: a blocked fp32 GEMM at a few GFLOP/s per core.
//
// Reference arithmetic restated (file:line in /root/reference and the installed transformers it delegates to):
//   visrep_vit_forward_cpu        clip_encoder.py:39-51 / dinov2_encoder.py:42-54 / siglip_encoder.py:40-52 -> HF CLIPVisionModel /
//                                 Dinov2Model / SiglipVisionModel(output_hidden_states=True).hidden_states[k] (pre-LN ViT blocks)
//   visrep_ascore_maxcos_cpu      A_score/compute.py:12-15 (normalize_feat) and :54-72 (F.cosine_similarity -> max(dim=1) -> mean)
//   visrep_cscore_transfer_cpu    C_score/pck_train.py:24-29, pck_train_two.py:24-36, utils/utils_correspondence.py:297-337,345-382
//   visrep_pck_count_cpu          C_score/pck_train.py:101,149-163
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>

#include "visrep_internal.h"

namespace {

int clamp_threads(int threads) {
    if (threads <= 0) {
        const unsigned hc = std::thread::hardware_concurrency();
        threads = hc ? (int)std::min(hc, 32u) : 4;                // torch-style oversubscription does not pay on many-core hosts
    }
    return std::max(1, std::min(threads, 256));
}

// dynamic work queue: fn(i) for i in [0, n)
void parallel_for(long n, int threads, const std::function<void(long)>& fn) {
    threads = (int)std::min<long>(clamp_threads(threads), std::max<long>(n, 1));
    if (threads <= 1) {
        for (long i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<long> next(0);
    std::vector<std::thread> pool;
    pool.reserve(threads);
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&]() {
            for (;;) {
                const long i = next.fetch_add(1);
                if (i >= n) return;
                fn(i);
            }
        });
    for (auto& th : pool) th.join();
}

// C[M, N] (ldc) = A[M, K] (lda) * W[N, K]^T (ldw) + bias[N]: nn.Linear.  Rows of A are independent work items (blocks of 8 rows); the
// inner loop is a dot product over contiguous K that the compiler vectorises.  fp32 accumulation in eight partial sums (pairwise-ish).
void linear_rows(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, long M, int N, int K, int threads) {
    const long blocks = (M + 7) / 8;
    parallel_for(blocks, threads, [&](long b) {
        const long m0 = b * 8, m1 = std::min<long>(M, m0 + 8);
        for (int n = 0; n < N; ++n) {
            const float* w = W + (size_t)n * ldw;
            for (long m = m0; m < m1; ++m) {
                const float* a = A + (size_t)m * lda;
                float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int k = 0;
                for (; k + 8 <= K; k += 8)
                    for (int u = 0; u < 8; ++u) s[u] += a[k + u] * w[k + u];
                float t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
                for (; k < K; ++k) t += a[k] * w[k];
                C[(size_t)m * ldc + n] = t + (bias ? bias[n] : 0.f);
            }
        }
    });
}

void layernorm_rows_cpu(const float* x, const float* g, const float* b, float* y, long rows, int d, float eps, int threads) {
    parallel_for((rows + 63) / 64, threads, [&](long blk) {
        for (long r = blk * 64; r < std::min<long>(rows, blk * 64 + 64); ++r) {
            const float* xr = x + (size_t)r * d;
            float* yr = y + (size_t)r * d;
            double mean = 0.0;
            for (int i = 0; i < d; ++i) mean += xr[i];
            mean /= d;
            double var = 0.0;
            for (int i = 0; i < d; ++i) { const double t = xr[i] - mean; var += t * t; }
            var /= d;                                              // biased, like torch.nn.LayerNorm
            const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
            for (int i = 0; i < d; ++i) yr[i] = (xr[i] - mu) * rstd * g[i] + b[i];
        }
    });
}

float act_cpu(float x, int act) {
    switch (act) {
        case VISREP_ACT_QUICK_GELU: return x * (1.0f / (1.0f + expf(-1.702f * x)));                     // HF QuickGELUActivation
        case VISREP_ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));             // nn.GELU()
        case VISREP_ACT_GELU_TANH: return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
        default: return x;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ ViT tower
extern "C" int visrep_vit_forward_cpu(const visrep_vit_config* c, const visrep_vit_weights* w, const float* pixels, float* hidden, int B, int n_layers,
                                      int threads) {
    if (!c || !w || !pixels || !hidden) return visrep_set_error(VISREP_ERR_ARG, "vit_forward_cpu: null pointer");
    if (B <= 0) return 0;
    if (n_layers < 0 || n_layers > c->layers) return visrep_set_error(VISREP_ERR_ARG, "vit_forward_cpu: n_layers out of range");
    if (c->heads <= 0 || c->d % c->heads) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward_cpu: d must be a multiple of heads");
    const int grid = c->image_size / c->patch, P = grid * grid, T = c->tokens, d = c->d, p = c->patch, S = c->image_size;
    if (grid * c->patch != S || P + (c->has_cls ? 1 : 0) != T) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward_cpu: tokens != grid^2 + cls");
    const int kp = 3 * p * p, kpad = c->kpad > 0 ? c->kpad : kp;       // patch_w rows are kpad floats wide (zero padded past 3 p p)
    if (kpad < kp) return visrep_set_error(VISREP_ERR_SHAPE, "vit_forward_cpu: kpad < 3 * patch^2");
    const long M = (long)B * T;
    const int dh = d / c->heads, mlp = c->mlp;
    const float* pw = (const float*)w->patch_w;
    // ---- embeddings: Conv2d(3, d, k = s = patch) as im2col (column order (channel, ky, kx) = the conv weight's own flattening) + Linear
    std::vector<float> cols((size_t)B * P * kp);
    parallel_for((long)B * P, threads, [&](long i) {
        const int b = (int)(i / P), pi = (int)(i % P), gy = pi / grid, gx = pi % grid;
        float* o = cols.data() + (size_t)i * kp;
        for (int ch = 0; ch < 3; ++ch)
            for (int ky = 0; ky < p; ++ky)
                memcpy(o + (ch * p + ky) * p, pixels + (((size_t)b * 3 + ch) * S + gy * p + ky) * S + gx * p, sizeof(float) * p);
    });
    std::vector<float> emb((size_t)B * P * d);
    linear_rows(cols.data(), kp, pw, kpad, w->patch_b, emb.data(), d, (long)B * P, d, kp, threads);
    const int off = c->has_cls ? 1 : 0;
    parallel_for(M, threads, [&](long r) {
        const int b = (int)(r / T), t = (int)(r % T);
        float* x = hidden + (size_t)r * d;
        const float* pos = w->pos + (size_t)t * d;
        if (c->has_cls && t == 0) {
            for (int i = 0; i < d; ++i) x[i] = w->cls[i] + pos[i];
        } else {
            const float* e = emb.data() + ((size_t)b * P + (t - off)) * d;
            for (int i = 0; i < d; ++i) x[i] = e[i] + pos[i];
        }
    });
    std::vector<float>().swap(cols);
    std::vector<float>().swap(emb);
    if (c->pre_ln) layernorm_rows_cpu(hidden, w->pre_ln_g, w->pre_ln_b, hidden, M, d, c->eps, threads);
    std::vector<float> h((size_t)M * d), qkv((size_t)M * 3 * d), att((size_t)M * d), mid((size_t)M * mlp), tmp((size_t)M * d);
    const float scale = 1.0f / sqrtf((float)dh);
    for (int l = 0; l < n_layers; ++l) {
        const visrep_vit_layer& L = w->layers[l];
        layernorm_rows_cpu(hidden, L.ln1_g, L.ln1_b, h.data(), M, d, c->eps, threads);
        linear_rows(h.data(), d, (const float*)L.wqkv, d, L.bqkv, qkv.data(), 3 * d, M, 3 * d, d, threads);
        // softmax(q k^T / sqrt(dh)) v per (image, head, query row), fp32 with a max-subtracted exponent like torch.softmax
        parallel_for((long)B * c->heads, threads, [&](long bh) {
            const int b = (int)(bh / c->heads), hd = (int)(bh % c->heads);
            std::vector<float> sc(T);
            for (int tq = 0; tq < T; ++tq) {
                const float* q = qkv.data() + ((size_t)b * T + tq) * 3 * d + hd * dh;
                float mx = -INFINITY;
                for (int tk = 0; tk < T; ++tk) {
                    const float* k = qkv.data() + ((size_t)b * T + tk) * 3 * d + d + hd * dh;
                    float s = 0.f;
                    for (int i = 0; i < dh; ++i) s += q[i] * k[i];
                    sc[tk] = s * scale;
                    mx = std::max(mx, sc[tk]);
                }
                float den = 0.f;
                for (int tk = 0; tk < T; ++tk) { sc[tk] = expf(sc[tk] - mx); den += sc[tk]; }
                float* o = att.data() + ((size_t)b * T + tq) * d + hd * dh;
                for (int i = 0; i < dh; ++i) o[i] = 0.f;
                for (int tk = 0; tk < T; ++tk) {
                    const float pr = sc[tk] / den;
                    const float* v = qkv.data() + ((size_t)b * T + tk) * 3 * d + 2 * d + hd * dh;
                    for (int i = 0; i < dh; ++i) o[i] += pr * v[i];
                }
            }
        });
        linear_rows(att.data(), d, (const float*)L.wo, d, L.bo, tmp.data(), d, M, d, d, threads);
        parallel_for((M + 63) / 64, threads, [&](long blk) {
            for (long r = blk * 64; r < std::min<long>(M, blk * 64 + 64); ++r)
                for (int i = 0; i < d; ++i) hidden[(size_t)r * d + i] += (L.ls1 ? L.ls1[i] : 1.f) * tmp[(size_t)r * d + i];
        });
        layernorm_rows_cpu(hidden, L.ln2_g, L.ln2_b, h.data(), M, d, c->eps, threads);
        linear_rows(h.data(), d, (const float*)L.w1, d, L.b1, mid.data(), mlp, M, mlp, d, threads);
        const int act = c->act;
        parallel_for((M + 15) / 16, threads, [&](long blk) {
            for (size_t i = (size_t)blk * 16 * mlp; i < (size_t)std::min<long>(M, blk * 16 + 16) * mlp; ++i) mid[i] = act_cpu(mid[i], act);
        });
        linear_rows(mid.data(), mlp, (const float*)L.w2, mlp, L.b2, tmp.data(), d, M, d, mlp, threads);
        parallel_for((M + 63) / 64, threads, [&](long blk) {
            for (long r = blk * 64; r < std::min<long>(M, blk * 64 + 64); ++r)
                for (int i = 0; i < d; ++i) hidden[(size_t)r * d + i] += (L.ls2 ? L.ls2[i] : 1.f) * tmp[(size_t)r * d + i];
        });
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ A score
extern "C" int visrep_ascore_maxcos_cpu(const float* other, const float* ref, int n_img, int Nt, int Nr, int D, float* scores, int threads) {
    if (n_img <= 0) return 0;
    if (!other || !ref || !scores) return visrep_set_error(VISREP_ERR_ARG, "ascore_cpu: null pointer");
    if (Nt <= 0 || Nr <= 0 || D <= 0) return visrep_set_error(VISREP_ERR_SHAPE, "ascore_cpu: empty tensor");
    // both normalisations of a row (compute.py:12-15: x / (|x| + 1e-10); cosine_similarity: / max(|x'|, 1e-8)) as one factor
    auto row_factor = [D](const float* x) {
        double ss = 0.0;
        for (int i = 0; i < D; ++i) ss += (double)x[i] * x[i];
        const float n = (float)sqrt(ss), f1 = 1.0f / (n + 1e-10f);
        return f1 / std::max(n * f1, 1e-8f);
    };
    std::vector<float> row_max((size_t)n_img * Nt);
    parallel_for((long)n_img * ((Nt + 15) / 16), threads, [&](long item) {
        const int tb = (Nt + 15) / 16, img = (int)(item / tb), t0 = (int)(item % tb) * 16, t1 = std::min(Nt, t0 + 16);
        const float* R = ref + (size_t)img * Nr * D;
        std::vector<float> rf(Nr);
        for (int s = 0; s < Nr; ++s) rf[s] = row_factor(R + (size_t)s * D);
        for (int t = t0; t < t1; ++t) {
            const float* o = other + ((size_t)img * Nt + t) * D;
            const float of = row_factor(o);
            float best = -INFINITY;
            for (int s = 0; s < Nr; ++s) {
                const float* r = R + (size_t)s * D;
                float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int k = 0;
                for (; k + 8 <= D; k += 8)
                    for (int u = 0; u < 8; ++u) acc[u] += o[k + u] * r[k + u];
                float dot = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
                for (; k < D; ++k) dot += o[k] * r[k];
                best = std::max(best, dot * of * rf[s]);
            }
            row_max[(size_t)img * Nt + t] = best;
        }
    });
    for (int img = 0; img < n_img; ++img) {
        double s = 0.0;
        for (int t = 0; t < Nt; ++t) s += row_max[(size_t)img * Nt + t];
        scores[img] = (float)(s / Nt);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ C score
extern "C" int visrep_cscore_transfer_cpu(const float* feats, const int* img1, const int* img2, const int* patch_idx, const int* nkp, const float* lin,
                                          float* xy, int n_pairs, int kmax, int P, int C, int split, int window, int soft_eval, float beta,
                                          float anno_stride, float anno_half, int layout, int threads) {
    if (n_pairs <= 0) return 0;
    if (!feats || !img1 || !img2 || !patch_idx || !nkp || !lin || !xy) return visrep_set_error(VISREP_ERR_ARG, "cscore_transfer_cpu: null pointer");
    if (P <= 0 || C <= 0 || kmax <= 0 || split < 0 || split >= C) return visrep_set_error(VISREP_ERR_SHAPE, "cscore_transfer_cpu: bad shape");
    const int PP = P * P;
    const size_t map = (size_t)PP * C;
    // descriptor of patch p of image im as a normalised C-vector (pck_train.py:24-29; two encoders: pck_train_two.py:24-36)
    auto descriptor = [&](int im, int pidx, float* out) {
        const float* base = feats + (size_t)im * map;
        for (int ch = 0; ch < C; ++ch) out[ch] = layout ? base[(size_t)pidx * C + ch] : base[(size_t)ch * PP + pidx];
        auto norm = [&](int c0, int c1) {
            double ss = 0.0;
            for (int ch = c0; ch < c1; ++ch) ss += (double)out[ch] * out[ch];
            const float f = 1.0f / ((float)sqrt(ss) + 1e-10f);
            for (int ch = c0; ch < c1; ++ch) out[ch] *= f;
        };
        if (split > 0) { norm(0, split); norm(split, C); }
        norm(0, C);
    };
    parallel_for(n_pairs, threads, [&](long z) {
        const int K = nkp[z];
        std::vector<float> tgt((size_t)PP * C), src(C), sim(PP), prob(PP);
        for (int q = 0; q < PP; ++q) descriptor(img2[z], q, tgt.data() + (size_t)q * C);
        for (int k = 0; k < kmax; ++k) {
            float* o = xy + ((size_t)z * kmax + k) * 2;
            if (k >= K) { o[0] = o[1] = 0.f; continue; }
            const int pidx = patch_idx[(size_t)z * kmax + k];
            if (pidx < 0 || pidx >= PP) { o[0] = o[1] = 0.f; continue; }
            descriptor(img1[z], pidx, src.data());
            int am = 0;
            for (int q = 0; q < PP; ++q) {
                const float* t = tgt.data() + (size_t)q * C;
                float s = 0.f;
                for (int ch = 0; ch < C; ++ch) s += src[ch] * t[ch];
                sim[q] = s;
                if (s > sim[am]) am = q;
            }
            float px, py;
            if (!soft_eval) {
                px = (float)(am % P); py = (float)(am / P);
            } else {
                // get_flow (utils_correspondence.py:297-337): entries outside the clamped window become ZERO (not -inf) and stay in the softmax
                if (window > 0) {
                    const int mx = am % P, my = am / P;
                    const int x0 = std::max(0, mx - window), x1 = std::min(P - 1, mx + window), y0 = std::max(0, my - window), y1 = std::min(P - 1, my + window);
                    for (int q = 0; q < PP; ++q) {
                        const int qx = q % P, qy = q / P;
                        if (qx < x0 || qx > x1 || qy < y0 || qy > y1) sim[q] = 0.f;
                    }
                } else if (window < 0) {
                    // "kernel soft-argmax" (apply_gaussian_kernel, utils_correspondence.py:278-295): a Gaussian of sigma = -window patches around the
                    // argmax target weights every similarity before the softmax (the reference hard-wires a 60 x 60 grid; any P here)
                    const float mx = (float)(am % P), my = (float)(am / P), two_s2 = 2.f * (float)(window * window);
                    for (int q = 0; q < PP; ++q) {
                        const float dx = (float)(q % P) - mx, dy = (float)(q / P) - my;
                        sim[q] *= expf(-(dx * dx + dy * dy) / two_s2);
                    }
                }
                float mxv = sim[0];
                for (int q = 1; q < PP; ++q) mxv = std::max(mxv, sim[q]);
                double den = 0.0;
                for (int q = 0; q < PP; ++q) { prob[q] = expf((sim[q] - mxv) / beta); den += prob[q]; }
                double gx = 0.0, gy = 0.0;
                for (int q = 0; q < PP; ++q) {
                    const double pr = prob[q] / den;
                    gx += pr * lin[q % P];
                    gy += pr * lin[q / P];
                }
                px = (float)((gx + 1.0) * (P - 1) / 2.0);
                py = (float)((gy + 1.0) * (P - 1) / 2.0);
                px = std::min(std::max(px, 0.f), (float)(P - 1));
                py = std::min(std::max(py, 0.f), (float)(P - 1));
            }
            o[0] = px * anno_stride + anno_half;
            o[1] = py * anno_stride + anno_half;
        }
    });
    return 0;
}

extern "C" int visrep_pck_count_cpu(const float* xy, const float* kps1, const float* kps2, const double* thresholds, const int* nkp, int n_pairs,
                                    int kmax, const float* alphas3, int* counts) {
    if (n_pairs <= 0) return 0;
    if (!xy || !kps1 || !kps2 || !thresholds || !nkp || !alphas3 || !counts) return visrep_set_error(VISREP_ERR_ARG, "pck_count_cpu: null pointer");
    for (int z = 0; z < n_pairs; ++z) {
        int hit[3] = {0, 0, 0}, vis = 0;
        for (int k = 0; k < nkp[z] && k < kmax; ++k) {
            const float* a = kps1 + ((size_t)z * kmax + k) * 3;
            const float* b = kps2 + ((size_t)z * kmax + k) * 3;
            if (!(a[2] * b[2] > 0.f)) continue;                   // vis = img1_kps[:, 2] * img2_kps[:, 2] > 0 (pck_train.py:101)
            ++vis;
            const float dx = b[0] - xy[((size_t)z * kmax + k) * 2], dy = b[1] - xy[((size_t)z * kmax + k) * 2 + 1];
            const float err = sqrtf(dy * dy + dx * dx);           // fp32 norm of the (y, x) difference (pck_train.py:149-152)
            for (int j = 0; j < 3; ++j)
                if ((double)err < (double)alphas3[j] * thresholds[z]) ++hit[j];       // float32 alpha x float64 bbox threshold (:157-160)
        }
        counts[z * 4 + 0] = hit[0]; counts[z * 4 + 1] = hit[1]; counts[z * 4 + 2] = hit[2]; counts[z * 4 + 3] = vis;
    }
    return 0;
}
