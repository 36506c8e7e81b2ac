#!/usr/bin/env python
"""SD1.5 feature tower at the reference's working point (768x768 input, up_ft_index 0, t=261 -> [B, 576, 1280]):
images/s and per-stage times with HIP events.  Synthetic weights.  Usage: python tools/sd_bench.py [batch] [reps] [side] [SD_SPECS key]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import sd_engine as SE, sd_weights as SW  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
side = int(sys.argv[3]) if len(sys.argv) > 3 else 768
dev = torch.device("cuda:0")
sp = SW.SD_SPECS[sys.argv[4] if len(sys.argv) > 4 else "runwayml/stable-diffusion-v1-5"]
t0 = time.time()
eng = SE.SdEngine(sp, SW.synthetic_unet(sp.unet, 21, 1), SW.synthetic_vae(sp.vae, 22), dev)
print(f"weights + packing: {time.time() - t0:.1f}s", flush=True)
rs = np.random.RandomState(0)
img = torch.from_numpy(rs.uniform(-1, 1, (B, 3, side, side)).astype(np.float32)).to(dev)
pe = torch.from_numpy(rs.standard_normal((1, 77, sp.unet.cross_dim)).astype(np.float32))
eng.set_prompt(pe)
eng.set_timestep(261)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


ms_vae, (mom, h, w) = timed(lambda: eng.vae_moments(img), reps)
lat = torch.randn(B * h * w, 8, device=dev).to(torch.bfloat16)
ms_unet, ft = timed(lambda: eng.unet_features(lat, B, h, w), reps)
ms_all, out = timed(lambda: eng.forward(img, t=261), reps)
eng.graph = False
ms_eager, _ = timed(lambda: eng.forward(img, t=261), reps)
print(f"eager forward {ms_eager:.1f} ms vs graph {ms_all:.1f} ms")
print(f"B={B} side={side}: vae {ms_vae:.1f} ms, unet {ms_unet:.1f} ms, forward {ms_all:.1f} ms -> {B / ms_all * 1e3:.2f} img/s; "
      f"out {tuple(out.shape)} finite={bool(torch.isfinite(out.float()).all())} peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
