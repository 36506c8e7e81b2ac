"""A/B of the GEMM variants on the five ViT-L/14-336 batch-256 shapes (interleaved rounds, HIP events, median + min).
usage: gemm_v4_time.py [variants, default 2,4] [rounds] [epilogue set: all|plain]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine  # noqa: E402

dev = "cuda:0"
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,4").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
which = sys.argv[3] if len(sys.argv) > 3 else "all"
M, d, m = 256 * 577, 1024, 4096
Mh = 147456                                     # rows the 256x256 kernels' full rounds cover (the rest is a 128x128 tail launch)
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x = rn(M, d).to(torch.bfloat16)
hm = rn(M, m).to(torch.bfloat16)
w1 = (rn(m, d) * 0.02).to(torch.bfloat16)
w2 = (rn(d, m) * 0.02).to(torch.bfloat16)
wqk = (rn(2 * d, d) * 0.02).to(torch.bfloat16)
wo = (rn(d, d) * 0.02).to(torch.bfloat16)
b1 = rn(m)
o1 = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
o2 = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
of = torch.empty(M, d, dtype=torch.float32, device=dev)
oqk = torch.empty(M, 2 * d, dtype=torch.bfloat16, device=dev)
vt = torch.zeros(d, M + 128, dtype=torch.bfloat16, device=dev)
lib = _lib.load()
cases = {
    "vt   Mh x 1024 x 1024 (EPI_VT)": (lambda: engine.linear_vt(x[:Mh], wo, None, out=vt), 2.0 * Mh * d * d),
    "f32  Mh x 1024 x 4096 (EPI_F32)": (lambda: engine.gemm(hm[:Mh], w2, None, _lib.EPI_F32, out=of[:Mh]), 2.0 * Mh * d * m),
}
if which == "all":
    cases.update({
        "fc1  Mh x 4096 x 1024 (bias+QuickGELU)": (lambda: engine.gemm(x[:Mh], w1, b1, _lib.EPI_ACT, act="quick_gelu", out=o1[:Mh]), 2.0 * Mh * m * d),
        "fc2  Mh x 1024 x 4096 (bias+residual)": (lambda: engine.gemm(hm[:Mh], w2, None, _lib.EPI_RESID, resid=o2[:Mh], out=o2[:Mh]), 2.0 * Mh * m * d),
        "qk   Mh x 2048 x 1024 (bias)": (lambda: engine.gemm(x[:Mh], wqk, None, _lib.EPI_BIAS, out=oqk[:Mh]), 2.0 * Mh * 2 * d * d),
        "out  Mh x 1024 x 1024 (bias+residual)": (lambda: engine.gemm(x[:Mh], wo, None, _lib.EPI_RESID, resid=o2[:Mh], out=o2[:Mh]), 2.0 * Mh * d * d),
    })


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {(n, v): [] for n in cases for v in variants}
for r in range(rounds):
    for n, (fn, fl) in cases.items():
        for v in variants:
            lib.visrep_set_gemm_variant(v)
            res[(n, v)].append(t(fn))
for n, (fn, fl) in cases.items():
    line = f"{n:42s}"
    for v in variants:
        ts = sorted(res[(n, v)])
        med, mn = ts[len(ts) // 2], ts[0]
        line += f" | v{v}: {med:.4f} ms ({fl / med / 1e9:7.1f} TF) min {mn:.4f}"
    print(line, flush=True)
lib.visrep_set_gemm_variant(2)
