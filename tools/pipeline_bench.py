#!/usr/bin/env python
"""End-to-end throughput of the C-feature extraction path: JPEG files -> decode -> resize -> normalise -> tower
-> one .pt per image, for the host pipeline variants (serial decode like the reference, decode pool one batch ahead of the GPU,
resize + normalise on the device).  Usage: python tools/pipeline_bench.py [n_images] [feature]"""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd.C_score import extract_feature as EF  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
feature = sys.argv[2] if len(sys.argv) > 2 else "DINOv2"
root = tempfile.mkdtemp(prefix="visrep_pipe_")
src = os.path.join(root, "JPEGImages", "cat")
os.makedirs(src)
rs = np.random.RandomState(0)
base = rs.randint(0, 255, (375, 500, 3), dtype=np.uint8)
for i in range(n):
    Image.fromarray(np.roll(base, i, axis=1)).save(os.path.join(src, f"im{i:05d}.jpg"), quality=90)
EF.configure(feature, img_size=224, synthetic_weights=True, batch=64)
out = {"images": n, "feature": feature, "cores": os.cpu_count()}
warm = os.path.join(root, "warm", "JPEGImages", "cat")                 # untimed first pass: module load, workspace, first-touch
os.makedirs(warm)
for i in range(64):
    shutil.copy(os.path.join(src, f"im{i:05d}.jpg"), warm)
import contextlib
import io
with contextlib.redirect_stdout(io.StringIO()):
    EF.process_images(os.path.join(root, "warm", "JPEGImages"), os.path.join(root, "warm", "out"), workers=8)
for tag, workers, devpre in (("serial decode + host resize (reference order)", 1, False), ("decode pool x8, host resize", 8, False),
                             ("decode pool x32, host resize", 32, False), ("decode x1, device resize/normalise", 1, True),
                             ("decode pool x32, device resize/normalise", 32, True)):
    EF._state.device_preprocess = devpre
    dst = os.path.join(root, f"features_{workers}_{int(devpre)}")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        EF.process_images(os.path.join(root, "JPEGImages"), dst, workers=workers)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[tag] = {"images_per_s": round(n / dt, 1), "seconds": round(dt, 2)}
# tower alone on resident pixels, same batch size
px = torch.randn(64, 3, 224, 224).to(torch.bfloat16).cuda()
f = EF._state.dift.forward
f(px)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    f(px)
torch.cuda.synchronize()
out["tower only, batch 64 resident"] = {"images_per_s": round(8 * 64 / (time.perf_counter() - t0), 1)}
shutil.rmtree(root, ignore_errors=True)
print(json.dumps(out, indent=1))
