#!/usr/bin/env python
"""Feature-extraction throughput of every encoder setting of the paper on ONE MI355X (per-tower part):
ViT towers at batch 256, diffusion towers at the reference's working resolutions.  Synthetic weights (random values are
irrelevant to speed; drawn with torch for speed instead of the reproducible numpy stream).  Prints one JSON line per tower.
Usage: python tools/sweep_bench.py [names...]   (default: all)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import engine, sd_weights as SW, vit_weights as VW  # noqa: E402
from law_of_vision_representation_in_mllms_amd import dit_engine, sd3_engine, sd_engine  # noqa: E402

dev = torch.device("cuda:0")


def fast_synthetic(table, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in table:
        if name.endswith("bias"):
            out[name] = torch.randn(shape, generator=g) * 0.05
        elif "norm" in name.split(".")[-2]:
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            out[name] = torch.randn(shape, generator=g) * (1.0 / np.sqrt(int(np.prod(shape[1:]))))
    return out


SW._synthetic = fast_synthetic


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def vit(name, B, n_layers):
    spec = VW.SPECS[name]
    eng = engine.VitEngine(spec, VW.synthetic_weights(spec, seed=1, n_layers=n_layers), dev)
    px = torch.randn(B, 3, spec.image_size, spec.image_size, device=dev).to(torch.bfloat16)
    out = torch.empty(B, spec.tokens, spec.d, dtype=torch.bfloat16, device=dev)
    return (lambda: eng.forward(px, n_layers=n_layers, out=out)), B, f"{spec.image_size}px -> [{spec.tokens}, {spec.d}]"


def sd(name, B, side):
    sp = SW.SD_SPECS[name]
    eng = sd_engine.SdEngine(sp, SW.synthetic_unet(sp.unet, 21, 1), SW.synthetic_vae(sp.vae, 22), dev)
    img = torch.rand(B, 3, side, side, device=dev) * 2 - 1
    pe = torch.randn(1, 77, sp.unet.cross_dim)
    eng.set_prompt(pe)
    o = eng.forward(img, t=261)
    return (lambda: eng.forward(img, t=261)), B, f"{side}px -> {list(o.shape[1:])}"


def imsd(B, side):
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM.src.models.dift_imsd import IMSDFeaturizer
    os.environ["VISREP_SYNTHETIC_WEIGHTS"] = "1"
    f = IMSDFeaturizer(device=dev)
    img = torch.rand(B, 3, side, side, device=dev) * 2 - 1
    o = f.forward(img, "", t=261)
    return (lambda: f.forward(img, "", t=261)), B, f"{side}px -> {list(o.shape[1:])} (+ CLIP ViT-L/14 image embedding per image)"


def dit(B, side):
    sp = SW.DIT_SPECS["facebook/DiT-XL-2-512"]
    eng = dit_engine.DiTEngine(sp, SW.synthetic_dit(sp.core, 31), SW.synthetic_vae(sp.vae, 32), dev)
    img = torch.rand(B, 3, side, side, device=dev) * 2 - 1
    o = eng.forward(img, t=261)
    return (lambda: eng.forward(img, t=261)), B, f"{side}px -> {list(o.shape[1:])}"


def sd3(B, side):
    sp = SW.SD3_SPECS["stabilityai/stable-diffusion-3-medium-diffusers"]
    eng = sd3_engine.Sd3Engine(sp, SW.synthetic_sd3(sp.core, 61), SW.synthetic_vae(sp.vae, 62), dev)
    img = torch.rand(B, 3, side, side, device=dev) * 2 - 1
    pe = torch.zeros(1, 77 + 256, 4096)
    pe[:, :77, :2048] = torch.randn(1, 77, 2048)
    pooled = torch.randn(1, 2048)
    o = eng.forward(img, pe, t=1, pooled=pooled)
    return (lambda: eng.forward(img, pe, t=1, pooled=pooled)), B, f"{side}px -> {list(o.shape[1:])}"


TOWERS = {
    "CLIP336": lambda: vit("openai/clip-vit-large-patch14-336", 256, 23),
    "CLIP224 / OpenCLIP (same ViT-L/14-224 architecture)": lambda: vit("openai/clip-vit-large-patch14", 256, 23),
    "DINOv2-L@224": lambda: vit("facebook/dinov2-large", 256, 23),
    "SigLIP-B/16@224": lambda: vit("google/siglip-base-patch16-224", 256, 11),
    "SD1.5@768": lambda: sd("runwayml/stable-diffusion-v1-5", 4, 768),
    "SD2.1@768": lambda: sd("stabilityai/stable-diffusion-2-1", 4, 768),
    "SDXL@768": lambda: sd("stabilityai/stable-diffusion-xl-base-1.0", 4, 768),
    "SDim@768": lambda: imsd(4, 768),
    "DiT-XL/2@512": lambda: dit(8, 512),
    "SD3@1024": lambda: sd3(2, 1024),
}

want = sys.argv[1:] or list(TOWERS)
for name in TOWERS:
    if not any(w.lower() in name.lower() for w in want):
        continue
    t0 = time.time()
    fn, B, desc = TOWERS[name]()
    setup = time.time() - t0
    ms = timed(fn, 3)
    print(json.dumps({"tower": name, "batch": B, "ms_per_batch": round(ms, 2), "images_per_s": round(B / ms * 1e3, 1), "io": desc,
                      "setup_s": round(setup, 1), "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
    del fn
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
