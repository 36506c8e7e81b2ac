#!/usr/bin/env python
"""Headline benchmark: CLIP ViT-L/14-336 vision-tower feature extraction (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (patch-embed -> pre-LN -> 23 encoder layers -> hidden_states[-2], CLS dropped) over
one batch of 256 synthetic images per GPU, inputs already resident in HBM.  Images are independent units, so the path
shards over ranks with no data-path collective (weak scaling); value = images of ALL ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel (the bf16 MFMA GEMM, fc1 shape): algorithmic FLOP per launch / its average launch
                duration, timed with HIP events on the launch stream inside this run; peak = 2.5 PFLOP/s dense bf16.
  cpu_baseline  the CPU oracle (fp32 restatement of the reference's HF arithmetic, oracle/vit.py) on a bounded sample of
                the same workload on this host's cores (rank 0, N=1 only).  Reported baseline, not the target.
  sweep         the second half of BASELINE.json's metric - "A+C score wall-clock 13 encoders" (configs[4]): every setting of
                policy/fit.py:20 through towers -> mm_projector -> A score and towers -> feature bank -> C score
                (law_of_vision_representation_in_mllms_amd/sweep.py), images sharded over the ranks (STRONG scaling: fixed total work).
                Default (`--sweep full`): the reference's 100 images per encoder for A and the SPair-71k-size set for C (1,800
                images, 12,234 pairs per setting); `--sweep reduced` = a 1/10-size C set, `--sweep off` skips it.  It never changes
                the headline `value`.
  scores        the two score kernels at their single-GPU sizes (BASELINE.json configs[2] / [3]), timed in this run with HIP events:
                A score (bf16 MFMA Gram + row max + mean) at Nt = 576 / 256 / 196 target tokens against the 576- and 256-token
                references, C score (exact-fp32 MFMA keypoint transfer + PCK count) for 12,234 pairs at P = 16 / 24, each with its
                fraction of the roof that bounds it.
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver stack (set before HIP starts)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = "openai/clip-vit-large-patch14-336"
BATCH = 256
N_LAYERS = 23                     # select_layer = -2: the 24th layer is never needed
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0             # HBM3E (MI355X_MICROARCH.md)
# roofline.traffic = HBM bytes per fc1 launch.  Hardware counters cannot be read from inside this process (rocprofv3 owns them), so the
# figure is a CITATION of a committed PMC pass over this kernel at this shape, carried with its provenance: profiles/fc1_traffic.json
# (written by tools/summarize_pmc.py from `rocprofv3 --pmc` runs: FETCH_SIZE and WRITE_SIZE in separate passes, units and the gfx950
# FETCH correction as MI355X_MICROARCH.md "HBM" prescribes).  No matching record (other variant / batch, or no file) -> traffic is null.
FC1_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "fc1_traffic.json")


def fc1_traffic(variant, batch):
    try:
        with open(FC1_TRAFFIC_FILE) as fh:
            recs = json.load(fh)["records"]
    except (OSError, ValueError, KeyError):
        return None, None
    for r in recs:
        if r.get("gemm_variant") == variant and r.get("batch") == batch:
            return float(r["hbm_bytes_per_launch"]), {k: r[k] for k in ("source", "method", "collected", "kernel") if k in r}
    return None, None


def flops_per_image(spec, n_layers):
    T, d, m = spec.tokens, spec.d, spec.mlp
    per_layer = 2 * T * d * 3 * d + 2 * T * d * d + 4 * T * T * d + 4 * T * d * m
    patch = 2 * spec.num_patches * (3 * spec.patch ** 2) * d
    return n_layers * per_layer + patch


def timed_steps(step, steps, warmup, dist=None, device=None):
    """The measurement protocol: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by device-sync + barrier + device-sync on
    both sides; returns (seconds = MAX over ranks, last step's result).  `device` None = host-only stepping (the gloo test of this
    function); with a device, torch.cuda.synchronize fences the HIP stream."""
    def sync():
        if device is not None:
            torch.cuda.synchronize(device)

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()
    result = None
    gc.collect()                              # the interpreter's cyclic collector stays out of the timed region (a full pass over the process'
    gc_was_on = gc.isenabled()                # tensors and weight dictionaries takes ~35 ms of host time here); collected before the warm-up steps so
    gc.disable()                              # that the timed steps follow them without an idle gap
    try:
        for _ in range(warmup):
            result = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            result = step()
        fence()
        dt = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, result


PEAK_F32_MFMA_TFLOPS = 157.3      # exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md


def score_extras(dev, n_a=256, n_img=1800, n_pairs=12234):
    """The A-score and C-score kernels at the SURVEY §8(d) sizes, HIP events on the launch stream (the library launches on torch's
    current stream).  A: [n, Nt, 4096] bf16 targets against clip336 [n, 576, 4096] and clip224 [n, 256, 4096] (row scales precomputed,
    as the sweep does): 2 Nt (576 + 256) 4096 flop per image.  C: DINOv2-L-shaped position-major bank [1800, P^2, 1024] fp32, K ~ U{3..20}
    key points per pair: 2 * 32 padded rows * P^2 * C flop per pair on the exact-fp32 matrix pipe."""
    from law_of_vision_representation_in_mllms_amd import ascore_ops, cscore_ops

    def ev_time(fn, reps=5, warm=2, blocks=3):
        """best of `blocks` x `reps` launches (HIP events): these extras run after seconds of sustained matrix work, and the MFMA-bound ones
        are clock-sensitive (profiles/round3_final_kernel_stats.md)"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(dev)
        best = float("inf")
        for _ in range(blocks):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
        return best

    out = {}
    g = torch.Generator(device=dev).manual_seed(3)
    r336 = torch.randn(n_a, 576, 4096, device=dev, generator=g).to(torch.bfloat16)
    r224 = torch.randn(n_a, 256, 4096, device=dev, generator=g).to(torch.bfloat16)
    s336, s224 = ascore_ops.row_scales(r336), ascore_ops.row_scales(r224)
    for Nt in (576, 256, 196):
        o = torch.randn(n_a, Nt, 4096, device=dev, generator=g).to(torch.bfloat16)
        so = ascore_ops.row_scales(o)
        sec = ev_time(lambda: (ascore_ops.max_cos_mean(o, r336, so, s336), ascore_ops.max_cos_mean(o, r224, so, s224)))
        tf = 2.0 * Nt * 832 * 4096 * n_a / sec / 1e12
        gbs = (2 * Nt + 832) * 4096 * 2.0 * n_a / sec / 1e9          # every operand row is read once per launch (no reuse across images)
        mf, hf = tf / PEAK_BF16_TFLOPS, gbs / PEAK_HBM_GBS
        # the roof the launch sits closer to names the bound: at Nt <= 256 the kernel streams its operands at ~5.3 TB/s
        # (0.66 of the 8 TB/s peak, 0.84 of the 6.3 TB/s the guide calls achievable) while the matrix pipe is at ~0.3: HBM-bound there
        out[f"ascore_Nt{Nt}"] = {"ms": round(sec * 1e3, 3), "images": n_a, "images_per_s": round(n_a / sec, 1), "tflops": round(tf, 1),
                                 "bound": "hbm" if hf > mf else "mfma (bf16)", "frac": round(max(hf, mf), 4), "mfma_frac": round(mf, 4),
                                 "operand_GB_per_s": round(gbs, 1), "hbm_frac": round(hf, 4),
                                 "hbm_frac_of_achievable_6300": round(gbs / 6300.0, 4)}
        del o, so
    del r336, r224, s336, s224
    rs = np.random.RandomState(5)
    for P in (16, 24):
        C_ = 1024
        bank = torch.randn(n_img, P * P, C_, device=dev, generator=g)
        i1 = torch.from_numpy(rs.randint(0, n_img, n_pairs).astype(np.int32))
        i2 = torch.from_numpy(rs.randint(0, n_img, n_pairs).astype(np.int32))
        nkp = torch.from_numpy(rs.randint(3, 21, n_pairs).astype(np.int32))
        idx = torch.from_numpy(rs.randint(0, P * P, (n_pairs, 20)).astype(np.int32))
        kps = torch.rand(n_pairs, 20, 3) * 839
        kps[:, :, 2] = 1
        thr = torch.from_numpy(rs.uniform(150, 700, n_pairs))
        i1, i2, nkp, idx, kps, thr = (t.to(dev) for t in (i1, i2, nkp, idx, kps, thr))
        packed = cscore_ops.packed_rows_on(dev, i1, i2, idx, nkp)          # host-side packing of a static pair list: done once, outside the timed launches
        sec = ev_time(lambda: cscore_ops.pck_counts(cscore_ops.transfer(bank, i1, i2, idx, nkp, P, layout="pc", packed=packed), kps, kps, thr, nkp))
        tiles, rows_used = int(packed[1].shape[0]), int(nkp.sum().item())
        tf = 2.0 * 32 * tiles * P * P * C_ / sec / 1e12                    # launched MFMA work: 32-row tiles (key points of several pairs per tile)
        tf_useful = 2.0 * rows_used * P * P * C_ / sec / 1e12               # 2 K P^2 C per pair: the key-point rows that carry data
        out[f"cscore_P{P}"] = {"ms": round(sec * 1e3, 3), "pairs": n_pairs, "pairs_per_s": round(n_pairs / sec, 1), "tiles": tiles,
                               "tile_fill": round(rows_used / (32.0 * tiles), 3), "tflops_fp32": round(tf, 1), "tflops_fp32_useful": round(tf_useful, 1),
                               "bound": "mfma (exact fp32)", "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                               "frac_useful": round(tf_useful / PEAK_F32_MFMA_TFLOPS, 4),
                               "unique_bank_GB_per_s": round(n_img * P * P * C_ * 4 / sec / 1e9, 1)}
        del bank
    torch.cuda.empty_cache()
    return out


def fp32_tower_extra(dev, spec, weights, batch=113):
    """The same tower in the reference's C-path precision (fp32: C_score/extract_feature.py:36-45), one launch of the sweep's size (113 images =
    one chunk: 3.98 rounds of the N = 1024 GEMMs' tiles; until round 4's second half: 64 images = 2.25 rounds): the default route (projections
    as split-bf16 GEMMs on the bf16 matrix pipe where the shapes allow) and the exact-fp32 MFMA route beside it."""
    from law_of_vision_representation_in_mllms_amd import engine
    px = torch.randn(batch, 3, spec.image_size, spec.image_size, device=dev)
    T, d, m = spec.tokens, spec.d, spec.mlp
    fl = N_LAYERS * (2 * T * d * 3 * d + 2 * T * d * d + 4 * T * T * d + 4 * T * d * m) * batch
    out = {"batch": batch, "default_products": engine.DEFAULT_SPLIT_PRODUCTS, "sweep_products": engine.THROUGHPUT_SPLIT_PRODUCTS}
    for route, products in (("split", engine.THROUGHPUT_SPLIT_PRODUCTS), ("split", engine.DEFAULT_SPLIT_PRODUCTS), ("native", None)):
        eng = engine.VitEngineF32(spec, weights, dev, gemm=route, products=products)
        for _ in range(2):
            eng.forward(px, n_layers=N_LAYERS)
        torch.cuda.synchronize(dev)
        sec = float("inf")
        for _ in range(3):                                                # best of three forwards (clock-sensitive, see score_extras)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.forward(px, n_layers=N_LAYERS)
            e1.record()
            torch.cuda.synchronize(dev)
            sec = min(sec, e0.elapsed_time(e1) * 1e-3)
        ent = {"ms": round(sec * 1e3, 1), "images_per_s": round(batch / sec, 1), "fp32_equivalent_tflops": round(fl / sec / 1e12, 1)}
        if eng.gemm == "split":      # the pipe it runs on is the bf16 one: `products` bf16 plane-pair products per fp32 product
            ent.update(products=eng.products, bf16_tflops=round(eng.products * fl / sec / 1e12, 1), bound="mfma (bf16)",
                       frac=round(eng.products * fl / sec / 1e12 / PEAK_BF16_TFLOPS, 4))
        else:
            ent.update(bound="mfma (exact fp32)", frac=round(fl / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))
        out[f"split{eng.products}" if eng.gemm == "split" else "native"] = ent
        del eng
    torch.cuda.empty_cache()
    return out


def aggregate_value(world, per_rank_units, steps, seconds):
    """Whole-job throughput: units of ALL ranks (weak scaling: every rank steps over its own batch) / max-over-ranks time."""
    return world * per_rank_units * steps / seconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=8)                 # 8 of the same images
    ap.add_argument("--sweep", default="full", choices=["off", "reduced", "full"])
    ap.add_argument("--sweep-precision", default="reference", choices=["reference", "bf16", "fp32"])   # reference: A leg bf16, C leg per the reference's scripts
    # split-bf16 product set of the sweep's fp32 ViT engines.  Round 6: `wall_s` is the fp32-EQUIVALENT sweep (six products over three-plane
    # operands - what stands in for the reference's true-fp32 C-leg towers, C_score/extract_feature.py:36-50); the throughput set (three products
    # over two-plane operands: 16 significand bits, narrower than the reference's arithmetic) is timed beside it as `wall_s_fp32x3` and labelled.
    # The choice is recorded in sweep.fp32_products and in every fp32 setting's dtype label.
    ap.add_argument("--sweep-fp32-products", type=int, default=6, choices=[3, 4, 6])
    ap.add_argument("--sweep-alt-fp32-products", type=int, default=3, choices=[0, 3, 4, 6])   # the fp32 C legs once more on this set ("wall_s_fp32x3"); 0 = off
    ap.add_argument("--no-scores", action="store_true")
    ap.add_argument("--gemm-variant", type=int, default=int(os.environ.get("VISREP_GEMM_VARIANT", "5")), choices=[1, 2, 5, 6, 7])   # 6 / 7: the round-6 duo kernel (A/B only, profiles/round6_gemm.md)
    args = ap.parse_args()

    torch.set_num_threads(min(32, os.cpu_count() or 1))   # host-side weight packing: torch's default (128 here) thrashes, see cpu_baseline
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the scoring path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)       # RCCL

    from law_of_vision_representation_in_mllms_amd import _lib, engine
    from law_of_vision_representation_in_mllms_amd import vit_weights as VW

    _lib.load().visrep_set_gemm_variant(args.gemm_variant)              # per-thread knob: this (the launching) thread
    spec = VW.SPECS[MODEL]
    weights = VW.synthetic_weights(spec, seed=1, n_layers=N_LAYERS)      # same tower replica on every rank
    eng = engine.VitEngine(spec, weights, dev)
    B = args.batch
    rs = np.random.RandomState(2 + rank)
    px = torch.from_numpy(rs.standard_normal((B, 3, spec.image_size, spec.image_size)).astype(np.float32))
    px = px.to(torch.bfloat16).to(dev)                                   # inputs resident in HBM before timing
    out = torch.empty(B, spec.tokens, spec.d, dtype=torch.bfloat16, device=dev)

    def step():
        return eng.forward(px, n_layers=N_LAYERS, out=out)[:, 1:]        # feature_select 'patch' (clip_encoder.py:31-32)

    dt, feats = timed_steps(step, args.steps, args.warmup, dist, dev)
    assert torch.isfinite(feats.float()).all()

    ms_per_step = dt / args.steps * 1e3
    value = aggregate_value(world, B, args.steps, dt)
    fl_img = flops_per_image(spec, N_LAYERS)

    # ---- roofline of the dominant kernel, timed with HIP events on the launch stream
    roof, kern = None, {}
    if rank == 0:
        M, d, m = B * spec.tokens, spec.d, spec.mlp
        # Operands = what the timed forward feeds these launches: the hidden state entering the middle encoder layer and that layer's weights
        # (LayerNorm gamma folded as the engine folds it), the MLP activations fc1 produces from them.  (Until round 4 these were N(0, 1)
        # samples: the same kernels run ~6 % slower on them than inside the forward - the matrix pipe's clock follows its operands' toggle
        # rate - so the micro-benchmark disagreed with the rocprofv3 average of the forward's own launches.)
        LAYER = N_LAYERS // 2
        x = eng.forward(px, n_layers=LAYER).reshape(M, d).clone()
        L0 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in weights["layers"][LAYER].items()}
        if eng.fuse_ln:
            L0["w1"] = L0["w1"].float() * L0["ln2_g"].float()[None]
            L0["wqkv"] = L0["wqkv"].float() * L0["ln1_g"].float()[None]
        w1 = L0["w1"].to(dev).to(torch.bfloat16)
        w2 = L0["w2"].to(dev).to(torch.bfloat16)
        wqk = L0["wqkv"][: 2 * d].to(dev).to(torch.bfloat16).contiguous()
        wo = L0["wo"].to(dev).to(torch.bfloat16)
        b1 = L0["b1"].to(dev)

        def time_kernel(fn, reps=20, warm=5):
            gc_was_on = gc.isenabled()       # a cyclic-collector pass between e0 and the first launch (35 ms of host time, seen in a kernel trace as a
            gc.disable()                     # launch-free gap) once turned twenty 0.45-ms attention launches into "2.3 ms each".  Disabled, not collected
            try:                             # here: a collection's 35 idle milliseconds drop the clock and cost the next launches ~10 %
                for _ in range(warm):        # steady state: the first launches after an idle gap run at a lower clock
                    fn()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize(dev)
                return e0.elapsed_time(e1) / reps * 1e-3
            finally:
                if gc_was_on:
                    gc.enable()

        o1 = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
        hmlp = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
        o2 = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
        oqk = torch.empty(M, 2 * d, dtype=torch.bfloat16, device=dev)
        lib = _lib.load()
        sp = _lib.stream_ptr
        if eng.fuse_ln:
            # the launches of the default forward: LayerNorm folded into fc1 / Q|K (per-row statistics applied in the epilogue),
            # residual GEMMs that also emit the next LayerNorm's statistics
            rt = torch.zeros((M + 127) // 128 * 128 + 8, 2, dtype=torch.float32, device=dev)
            part = torch.empty(M, d // 64, 2, dtype=torch.float32, device=dev)
            _lib.check(lib.visrep_layernorm_stats(_lib.ptr(x), d, _lib.ptr(rt), M, d, 1e-5, sp()), "stats")
            s1, sqk = w1.float().sum(1).contiguous(), wqk.float().sum(1).contiguous()
            bqk = torch.zeros(2 * d, dtype=torch.float32, device=dev)

            def fc1():
                _lib.check(lib.visrep_gemm_bf16_ln(_lib.ptr(x), d, _lib.ptr(w1), d, _lib.ptr(b1), _lib.ptr(rt), _lib.ptr(s1), _lib.ptr(o1), m, M, m, d,
                                                   _lib.EPI_ACT, _lib.ACT["quick_gelu"], sp()), "gemm_ln")

            def qk():
                _lib.check(lib.visrep_gemm_bf16_ln(_lib.ptr(x), d, _lib.ptr(wqk), d, _lib.ptr(bqk), _lib.ptr(rt), _lib.ptr(sqk), _lib.ptr(oqk), 2 * d, M,
                                                   2 * d, d, _lib.EPI_BIAS, 0, sp()), "gemm_ln")

            def resid(a, w, K):
                return lambda: _lib.check(lib.visrep_gemm_bf16_resid_stats(_lib.ptr(a), K, _lib.ptr(w), K, None, _lib.ptr(o2), d, M, d, K, _lib.ptr(o2), None,
                                                                           _lib.ptr(rt), _lib.ptr(part), 1e-5, sp()), "gemm_resid_stats")
            fc2, out_proj = resid(hmlp, w2, m), resid(x, wo, d)
        else:
            fc1 = lambda: engine.gemm(x, w1, b1, _lib.EPI_ACT, act="quick_gelu", out=o1)
            qk = lambda: engine.gemm(x, wqk, None, _lib.EPI_BIAS, out=oqk)
            fc2 = lambda: engine.gemm(hmlp, w2, None, _lib.EPI_RESID, resid=o2, out=o2)
            out_proj = lambda: engine.gemm(x, wo, None, _lib.EPI_RESID, resid=o2, out=o2)
        fc1()
        hmlp.copy_(o1)                                                   # fc2's operand: QuickGELU(fc1) of the same rows
        shapes = {
            "fc1 (M x 4096 x 1024, bias+QuickGELU)": (fc1, 2.0 * M * m * d),
            "fc2 (M x 1024 x 4096, bias+residual)": (fc2, 2.0 * M * m * d),
            "qk  (M x 2048 x 1024, bias)": (qk, 2.0 * M * 2 * d * d),
            "out (M x 1024 x 1024, bias+residual)": (out_proj, 2.0 * M * d * d),
        }
        for name, (fn, fl) in shapes.items():
            sec = time_kernel(fn)
            kern[name] = {"ms": round(sec * 1e3, 4), "tflops": round(fl / sec / 1e12, 1)}
        # attention (its own kernel): 4*T*T*d flop per image per layer
        # the launch of the default forward: Q pre-scaled by head_dim^-0.5 * log2(e) in the projection weights (engine.VitEngine), scale <= 0
        # operands: the Q | K projection of the same hidden state (what the qk launch above left in oqk), V from the layer's V weights
        qk()
        qk_f = oqk.float()
        if eng.q_prescaled:
            qk_f[:, :d] *= 0.125 * 1.4426950408889634
        qk_act = qk_f.to(torch.bfloat16)
        del qk_f
        wo = L0["wqkv"][2 * d:].to(dev).to(torch.bfloat16).contiguous()  # (only the attention operands below use it from here on)
        T = spec.tokens
        if getattr(eng, "_q_mode", 0) >= 2 and spec.has_cls and engine.mhsa_cls_supported(T):      # what visrep_vit_forward launches for this tower
            vt = engine.gemm_rows(x, T - 1, T, 1, B * (T - 1), wo, None, epilogue=_lib.EPI_VT)
            vcls = engine.gemm_rows(x, 1, T, 0, B, wo, None)
            sec = time_kernel(lambda: engine.mhsa_cls(qk_act, vt, vcls, B, T, spec.heads))
            which = "attn_fwd_cls (image-aligned key tiles, pre-scaled Q)"
        else:
            vt = engine.linear_vt(x, wo, None)
            sec = time_kernel(lambda: engine.mhsa(qk_act, vt, B, T, spec.heads, 0.0 if eng.q_prescaled else 0.125))
            which = "attn_fwd<1, pre-scaled Q>" if eng.q_prescaled else "attn_fwd<1>"
        kern["mhsa (577 tok, 16 heads)"] = {"ms": round(sec * 1e3, 4), "tflops": round(4.0 * B * T ** 2 * d / sec / 1e12, 1), "kernel": which}
        # The same launches IN THE LAYER'S LAUNCH MIX: Q|K -> attention -> out -> fc1 -> fc2, repeated like the 23 layers of the forward, with a HIP
        # event pair around every launch.  Twenty back-to-back launches of one matrix-bound kernel (the "ms" above) run 3-7 % slower than the same
        # kernel inside the forward - rocprofv3's average over the forward's launches (profiles/round4_final_kernel_stats.md, `fwd`) sits with the
        # in-mix figure, so that is the one the roofline object quotes; "ms" (back to back) stays beside it.
        attn_fn = (lambda: engine.mhsa_cls(qk_act, vt, vcls, B, T, spec.heads)) if which.startswith("attn_fwd_cls") else \
                  (lambda: engine.mhsa(qk_act, vt, B, T, spec.heads, 0.0 if eng.q_prescaled else 0.125))
        mix = [("qk  (M x 2048 x 1024, bias)", qk), ("mhsa (577 tok, 16 heads)", attn_fn), ("out (M x 1024 x 1024, bias+residual)", out_proj),
               ("fc1 (M x 4096 x 1024, bias+QuickGELU)", fc1), ("fc2 (M x 1024 x 4096, bias+residual)", fc2)]
        flops = {n: f for n, (_, f) in shapes.items()}
        flops["mhsa (577 tok, 16 heads)"] = 4.0 * B * T ** 2 * d
        for _ in range(2):
            for _, fn in mix:
                fn()
        torch.cuda.synchronize(dev)
        MIX_REPS = 12
        evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in mix] for _ in range(MIX_REPS)]
        for r in range(MIX_REPS):
            for i, (_, fn) in enumerate(mix):
                evs[r][i][0].record()
                fn()
                evs[r][i][1].record()
        torch.cuda.synchronize(dev)
        for i, (name, _) in enumerate(mix):
            msec = sum(evs[r][i][0].elapsed_time(evs[r][i][1]) for r in range(MIX_REPS)) / MIX_REPS
            kern[name]["ms_in_layer_mix"] = round(msec, 4)
            kern[name]["tflops_in_layer_mix"] = round(flops[name] / (msec * 1e-3) / 1e12, 1)
        del evs
        top = kern["fc1 (M x 4096 x 1024, bias+QuickGELU)"]
        # the 256x256 kernel on its own: the rows its full rounds cover (the dispatcher hands the last <= 256 rows to a 128x128 launch
        # pair) - this is the launch rocprofv3 lists as gemm_bf16_256<1>, so the two averages can be compared directly
        head = None
        try:
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            ntn, ntm = m // 256, (M + 255) // 256
            rounds = (ntm * ntn) // ncu
            m1 = rounds * ncu // ntn * 256
            if eng.fuse_ln and args.gemm_variant in (2, 5) and rounds >= 1 and (rounds * ncu) % ntn == 0 and 0 < m1 <= M:
                def fc1_head():
                    _lib.check(lib.visrep_gemm_bf16_ln(_lib.ptr(x), d, _lib.ptr(w1), d, _lib.ptr(b1), _lib.ptr(rt), _lib.ptr(s1), _lib.ptr(o1), m, m1, m, d,
                                                       _lib.EPI_ACT, _lib.ACT["quick_gelu"], sp()), "gemm_ln")
                hs = time_kernel(fc1_head)
                head = {"rows": m1, "ms": round(hs * 1e3, 4), "tflops": round(2.0 * m1 * m * d / hs / 1e12, 1)}
        except Exception as e:                                          # never let the extra line take the bench down
            head = {"error": str(e)[:200]}
        # ---- the comparable fc1 figure of BENCH_r01..r03: the same launch on N(0, 1) operands, twenty launches back to back.  The in-mix figure
        # above runs on the forward's own hidden state - produced by SYNTHETIC N(0, 0.02) weights, whose activations toggle fewer operand bits
        # than N(0, 1) samples (~6 % faster, clock-limited chip) - so both are on the line, labelled.
        n01 = None
        try:
            gx = torch.Generator(device=dev).manual_seed(12)
            xr = torch.randn(M, d, device=dev, generator=gx).to(torch.bfloat16)
            wr = (torch.randn(m, d, device=dev, generator=gx) * 0.02).to(torch.bfloat16)
            secr = time_kernel(lambda: engine.gemm(xr, wr, b1, _lib.EPI_ACT, act="quick_gelu", out=o1))
            n01 = {"ms": round(secr * 1e3, 4), "tflops": round(2.0 * M * m * d / secr / 1e12, 1), "frac": round(2.0 * M * m * d / secr / 1e12 / PEAK_BF16_TFLOPS, 4),
                   "operands": "X ~ N(0, 1), W ~ N(0, 0.02), bias + QuickGELU epilogue (no LayerNorm fold): the round-1..3 micro-benchmark"}
            del xr, wr
        except Exception as e:
            n01 = {"error": str(e)[:200]}
        # ---- the matrix pipe's practical ceiling on THIS box in THIS run: a free-running v_mfma_f32_16x16x32_bf16 stream on random register
        # operands (no memory traffic).  The chip's power management holds it near 1.9-2.0 PFLOP/s (DVFS), below the 2.5 PFLOP/s nominal peak
        # every `frac` on this line divides by - `frac_of_practical_roof` reads the same numbers against what the pipe can actually sustain.
        practical = None
        try:
            sink = torch.zeros(4, dtype=torch.int64, device=dev)   # [1] shader-clock cycles, [2] 100-MHz ticks of block 0's loop
            flop = C.c_double(0.0)
            best = 0.0
            for _ in range(2):                                   # ~15 ms each: long enough for the clock to settle under this load
                _lib.check(lib.visrep_debug_mfma_probe(20000, 1, _lib.ptr(sink), C.byref(flop), sp()), "mfma_probe")
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.visrep_debug_mfma_probe(20000, 1, _lib.ptr(sink), C.byref(flop), sp()), "mfma_probe")
                e1.record()
                torch.cuda.synchronize(dev)
                best = max(best, flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            cyc, ticks = sink[1].item(), sink[2].item()
            practical = {"tflops": round(best, 1), "frac_of_nominal": round(best / PEAK_BF16_TFLOPS, 4),
                         "clock_ghz": (round(cyc / (ticks * 10.0), 3) if ticks > 0 else None),
                         "clock_how": "s_memtime cycles / s_memrealtime (100 MHz) ticks of block 0's MFMA loop in the last launch: the shader clock the power management held under this stream (nominal peak is quoted at 2.4 GHz)",
                         "how": "visrep_debug_mfma_probe: 20000 x 32 v_mfma_f32_16x16x32_bf16 per wave, 8 waves per CU, random bf16 register operands with |x| in [0.25, 4), best of 3 launches after 2 warm-up launches, same process"}
        except Exception as e:
            practical = {"error": str(e)[:200]}
        # ---- board telemetry under the dominant kernel and under the timed forward: socket power against the board's cap and
        # the shader clock, sampled from amdgpu's sysfs files at ~200 Hz while the launches repeat for about a second each (telemetry.py).  A board
        # sitting at its cap with a clock below the 2.4 GHz the nominal peak is quoted at is what bounds `frac` here (profiles/round6_power.md).
        power = None
        try:
            from law_of_vision_representation_in_mllms_amd import telemetry
            def brief(t):
                if not t.get("available"):
                    return {"available": False}
                fw = t.get("firmware") or {}
                return {"power_w": t["power_w"]["mean"], "power_w_p95": t["power_w"]["p95"], "power_cap_w": t.get("power_cap_w"),
                        "sclk_mhz": t["sclk_mhz"]["mean"] if t.get("sclk_mhz") else None, "samples": t["samples"], "hz": t["hz"],
                        "frac_samples_ge_95pct_of_cap": t.get("frac_of_cap_samples_ge_95pct"), "ppt_residency": fw.get("ppt_residency"),
                        "throttle": ("power cap" if (t.get("frac_of_cap_samples_ge_95pct") or 0) >= 0.5 else "none seen")}
            power = {"fc1_loop": brief(telemetry.measure(fc1, 1.0, index=local)), "forward": brief(telemetry.measure(step, 1.5, index=local)),
                     "how": "hwmon power1_input / freq1_input of the card this process computes on, background thread, while the launch repeats; "
                            "sclk at the nominal peak is 2400 MHz; ppt_residency is decoded from the binary gpu_metrics table without an independent check: indicative only"}
        except Exception as e:
            power = {"error": f"{type(e).__name__}: {e}"[:200]}
        traffic, traffic_src = fc1_traffic(args.gemm_variant, B)
        roof = {"bound": "mfma", "kernel": {1: "gemm_bf16_128", 2: "gemm_bf16_256", 5: "gemm_bf16_256q"}.get(args.gemm_variant, "gemm_bf16_duo") + "<EPI_ACT> fc1",
                "achieved": top["tflops_in_layer_mix"], "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(top["tflops_in_layer_mix"] / PEAK_BF16_TFLOPS, 4),
                "timing": "HIP event pair around each fc1 dispatch (head launch + its 128x128 tail pair) inside the layer's launch mix, 12 layers' worth; "
                          "kernels.*.ms = 20 launches back to back",
                "achieved_basis": "fc1 inside the layer's launch mix on the timed forward's own operands (activations of synthetic N(0, 0.02) weights)",
                "achieved_back_to_back": top["tflops"], "frac_back_to_back": round(top["tflops"] / PEAK_BF16_TFLOPS, 4),
                "n01_back_to_back": n01,
                "practical_roof": practical,
                "power": power,
                "power_w": ((power or {}).get("fc1_loop") or {}).get("power_w"), "sclk_mhz": ((power or {}).get("fc1_loop") or {}).get("sclk_mhz"),
                "throttle": ((power or {}).get("fc1_loop") or {}).get("throttle"),
                "xcd_balance": dict(_lib.xcd_balance(), what="XCD-weighted tile split of the persistent 256x256 kernel: rel = measured time per round of tiles of each XCD relative to the mean (the XCDs run at their own clocks under the power limit); opt-in (VISREP_XCD_BALANCE=1), equal shares by default"),
                "frac_of_practical_roof": (round(top["tflops_in_layer_mix"] / practical["tflops"], 4) if practical and practical.get("tflops") else None),
                "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_provenance": traffic_src,
                "algorithmic_bytes_per_launch": 2.0 * (M * d + m * d + M * m),
                "flop_per_launch": 2.0 * M * m * d, "ms_per_launch": top["ms_in_layer_mix"], "dominant_kernel_only": head,
                "whole_forward": {"tflops": round(fl_img * value / world / 1e12, 1),
                                  "frac": round(fl_img * value / world / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "frac_of_practical_roof": (round(fl_img * value / world / 1e12 / practical["tflops"], 4) if practical and practical.get("tflops") else None),
                                  "gflop_per_image": round(fl_img / 1e9, 1)},
                "operands": f"the timed forward's hidden state entering encoder layer {LAYER} and that layer's weights: activations of SYNTHETIC N(0, 0.02) "
                            "weights (no checkpoint offline), not N(0, 1) samples - see n01_back_to_back for the comparable round-1..3 figure",
                "kernels": kern}
        del x, hmlp, o1, o2, oqk, qk_act, vt

    scores = None
    if rank == 0 and not args.no_scores:
        try:
            scores = score_extras(dev)
            scores["fp32_tower"] = fp32_tower_extra(dev, spec, weights)
        except Exception as e:                                              # never let the extra object take the bench down
            scores = {**(scores or {}), "error": f"{type(e).__name__}: {e}"[:300]}

    # ---- CPU baseline: the oracle on a bounded sample, host cores of this box (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_images > 0:
        from oracle import vit as OV
        n = args.cpu_images
        sample = px[:n].float().cpu()
        # Thread count: torch's CPU GEMMs do NOT scale to every logical CPU of this host class (measured with tools/cpu_probe.py on the
        # 256-logical-CPU MI355X box: 0.89 s/image at 32 threads, 1.5 at 64, 4.2 at 128; at 256 the 8-image sample did not finish in 20
        # minutes) - so the leg calibrates on one image over a few counts and reports the count it used as `cores`.
        ncpu = os.cpu_count() or 1
        best, best_t = None, None
        for nt in sorted({min(ncpu, c) for c in (16, 32, 64)}):
            torch.set_num_threads(nt)
            c0 = time.perf_counter()
            OV.tower_features(spec, weights, sample[:1], select_layer=N_LAYERS)
            c1 = time.perf_counter() - c0
            if best_t is None or c1 < best_t:
                best, best_t = nt, c1
            if c1 > 20:                                           # never let a pathological count eat the run
                break
        torch.set_num_threads(best)
        n = max(1, min(n, int(30.0 / max(best_t, 1e-3))))        # bounded sample: about 10-30 s of CPU work
        sample = sample[:n]
        c0 = time.perf_counter()
        ref = OV.tower_features(spec, weights, sample, select_layer=N_LAYERS)
        cdt = time.perf_counter() - c0
        got = feats[:n].float().cpu()
        rel = ((got - ref).norm() / ref.norm()).item()
        cpu = {"value": round(n / cdt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
               "host_logical_cpus": ncpu,
               "sample": f"{n} of the same 336x336 images, fp32, oracle/vit.py (torch CPU), 23 layers; thread count = fastest of 16/32/64 on one image",
               "gpu_vs_cpu_rel_l2": round(rel, 5)}

    # ---- BASELINE configs[0] at its stated size, once on the line: 32 images through the CLIP-L/14-224 tower, CPU float32, full
    # depth (hidden_states[-2]: 23 layers), through the product's HOST twin (visrep_vit_forward_cpu: the explicit device="cpu" engine, plumbing -
    # neither the headline nor the cpu_baseline above, which is the oracle on this run's own images)
    cfg0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            name0 = "openai/clip-vit-large-patch14"
            spec0 = VW.SPECS[name0]
            os.environ["VISREP_FAST_SYNTHETIC"] = "1"
            try:
                w0 = VW.synthetic_weights(spec0, seed=1, n_layers=N_LAYERS)
            finally:
                os.environ.pop("VISREP_FAST_SYNTHETIC", None)
            nthr = min(64, os.cpu_count() or 1)
            e0_ = engine.VitEngineCPU(spec0, w0, threads=nthr)
            px0 = torch.from_numpy(np.random.RandomState(9).standard_normal((32, 3, 224, 224)).astype(np.float32))
            c0 = time.perf_counter()
            e0_.forward(px0[:2], n_layers=N_LAYERS)                   # pages the weights in; also the estimate that bounds this leg
            est = (time.perf_counter() - c0) * 16
            n0 = 32 if est <= 90 else 8                               # never let a slow host eat the run: say so if the sample was cut
            c0 = time.perf_counter()
            f0 = e0_.forward(px0[:n0], n_layers=N_LAYERS)
            c1 = time.perf_counter() - c0
            cfg0 = {"config": "BASELINE configs[0]: CLIP ViT-L/14-224 vision_tower feature-extract, 32 images, CPU float32, 23 layers (hidden_states[-2])",
                    "images": n0, "seconds": round(c1, 2), "images_per_s": round(n0 / c1, 3), "threads": nthr, "host_logical_cpus": os.cpu_count(),
                    "path": "engine.VitEngineCPU -> visrep_vit_forward_cpu (csrc/host_twins.hip), synthetic weights, N(0,1) pixels", "finite": bool(torch.isfinite(f0).all())}
            del e0_, w0, f0
        except Exception as e:
            cfg0 = {"error": f"{type(e).__name__}: {e}"[:200]}

    def emit(sweep):
        if rank == 0:
            line = {
                "metric": "images/sec ViT-L/14@336 feature-extract", "value": round(value, 2), "unit": "images/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "CLIP ViT-L/14-336 vision_tower feature-extract (hidden_states[-2], 23 layers run), "
                                       f"batch {B} per GPU, random-init weights, N(0,1) pixels resident in HBM",
                           "global_batch": world * B, "tokens": spec.tokens, "parallelism": f"dp{world} (image-sharded, no collective)", "gemm_variant": args.gemm_variant},
                "roofline": roof, "cpu_baseline": cpu, "configs0_cpu": cfg0, "scores": scores, "sweep": sweep,
            }
            print(json.dumps(line), flush=True)

    # ---- the 13-setting A + C sweep (all ranks take part; images sharded rank::world).  It is an extra on the headline line and has the
    # path's only collectives: a rank that fails alone would leave the others waiting in an all-gather, so a watchdog bounds it - on
    # expiry rank 0 prints the line it already has (sweep = the timeout) and every rank leaves without the final barrier.
    sweep = None
    if args.sweep != "off":
        import threading
        # The headline line is printed AFTER the sweep (one JSON line per run), so a sweep that hangs must not take it down with it: the watchdog
        # bounds it.  One GPU: the full sweep takes ~80 s with its setup (900 s allowed).  world > 1 has never run on hardware (RCCL all_to_all under
        # persistent kernels): 420 s allowed - at N = 2 the sweep is ~30 s + ~25 s of setup - then rank 0 prints the line it has and every rank leaves.
        limit = float(os.environ.get("VISREP_SWEEP_LIMIT_S", "600" if args.sweep == "reduced" else ("900" if world == 1 else "420")))

        def bail():
            emit({"error": f"sweep did not finish within {limit:.0f} s", "size": args.sweep})
            sys.stdout.flush()
            os._exit(0)

        dog = threading.Timer(limit, bail)
        dog.daemon = True
        dog.start()
        try:
            del eng, px, out, feats
            torch.cuda.empty_cache()
            from law_of_vision_representation_in_mllms_amd import sweep as SW
            spair = SW.synthetic_spair() if args.sweep == "full" else SW.synthetic_spair(180, 1224)
            sweep = SW.run_sweep(SW.SETTINGS, 100, spair, dev, precision=args.sweep_precision, also_bf16=True, fp32_products=args.sweep_fp32_products,
                                 alt_fp32_products=(args.sweep_alt_fp32_products or None) if args.sweep_alt_fp32_products != args.sweep_fp32_products else None)
            sweep["size"] = args.sweep
        except Exception as e:                                           # the headline line must survive a sweep failure
            sweep = {"error": f"{type(e).__name__}: {e}"[:300]}
            if dist is not None:                                         # the other ranks may be inside a collective: do not join them
                dog.cancel()
                emit(sweep)
                sys.stdout.flush()
                os._exit(0)
        dog.cancel()

    emit(sweep)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
