#!/usr/bin/env python
"""A-score Gram kernels at the sweep's shapes: 128 x 128 tiles (variant 1) against the persistent ping-pong tiles (variant 2), HIP events,
row scales precomputed as the sweep does.  Usage: python tools/ascore_time.py [n_images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, ascore_ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda:0"
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(3)


def ev_time(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


only = int(os.environ.get("ASCORE_SHAPES", "0"))        # > 0: only the first shapes (PMC passes)
for Nt, Nr, D in ((576, 576, 4096), (576, 256, 4096), (256, 576, 4096), (256, 256, 4096), (196, 576, 4096), (196, 256, 4096), (729, 576, 4096), (576, 576, 1024))[: only or None]:
    o = torch.randn(n, Nt, D, device=dev, generator=g).to(torch.bfloat16)
    r = torch.randn(n, Nr, D, device=dev, generator=g).to(torch.bfloat16)
    so, sr = ascore_ops.row_scales(o), ascore_ops.row_scales(r)
    line = f"Nt={Nt:4d} Nr={Nr:4d} D={D}:"
    res = {}
    for v in (1, 2, 0):
        old = lib.visrep_set_ascore_variant(v)
        ms = ev_time(lambda: ascore_ops.max_cos_mean(o, r, so, sr))
        res[v] = ascore_ops.max_cos_mean(o, r, so, sr)
        lib.visrep_set_ascore_variant(old)
        tf = 2.0 * Nt * Nr * D * n / ms / 1e9
        line += f"  v{v} {ms:7.3f} ms {tf:6.0f} TF ({tf / 2500:.3f})"
    line += "  equal" if torch.equal(res[1], res[2]) else f"  DIFF {(res[1] - res[2]).abs().max().item():.3e}"
    print(line, flush=True)
    del o, r
