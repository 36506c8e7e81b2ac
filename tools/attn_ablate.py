"""Attribute the attention kernel's time: run the BASELINE shape (256 x 16 heads x 577 tokens) under each ablation library
(build.build_attn_ablation_lib).  Usage on the GPU box: python tools/attn_ablate.py   (libraries must be prebuilt in-tree)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from law_of_vision_representation_in_mllms_amd import engine
B, T, H, d = 256, 577, 16, 1024
M = B * T
qk = torch.randn(M, 2 * d, device="cuda").to(torch.bfloat16)
x = torch.randn(M, d, device="cuda").to(torch.bfloat16)
w = (torch.randn(d, d, device="cuda") * 0.03).to(torch.bfloat16)
vt = engine.linear_vt(x, w, None)
for _ in range(5): engine.mhsa(qk, vt, B, T, H, 0.125)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): engine.mhsa(qk, vt, B, T, H, 0.125)
e1.record(); torch.cuda.synchronize()
print("%%.4f" %% (e0.elapsed_time(e1) / 20))
''' % ROOT
names = {0: "full kernel", 1: "no v_exp", 2: "no P.V MFMA", 4: "no Q.K^T MFMA", 6: "no MFMA at all", 8: "no LDS fragment reads", 16: "no barrier / DMA per tile",
         24: "no LDS reads, no barrier/DMA", 7: "no MFMA, no exp", 31: "loop skeleton only"}
for lib in sorted(glob.glob(os.path.join(ROOT, "law_of_vision_representation_in_mllms_amd", "libvisrep_hip_attn*.so")),
                  key=lambda p: int(os.path.basename(p)[len("libvisrep_hip_attn"):-3])):
    mask = int(os.path.basename(lib)[len("libvisrep_hip_attn"):-3])
    env = dict(os.environ, VISREP_LIB=lib)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    ms = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-200:]
    print(f"mask {mask:2d}  {names.get(mask, ''):32s} {ms} ms", flush=True)
