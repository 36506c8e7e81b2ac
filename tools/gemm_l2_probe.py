"""Is the ping-pong GEMM waiting for memory?  The fc1 / fc2 / Q|K shapes with the real A operand and with every row of A aliased to row 0 (lda = 0:
the X panels then come out of L1 / L2, W and the output stream are unchanged), variant 5 and 2, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine  # noqa: E402

dev = "cuda:0"
Mh, d, m = 147456, 1024, 4096
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
lib = _lib.load()


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


for name, N, K, epi, act in (("fc1", m, d, _lib.EPI_ACT, "quick_gelu"), ("qk ", 2 * d, d, _lib.EPI_BIAS, "none"), ("fc2", d, m, _lib.EPI_BIAS, "none")):
    a = rn(Mh, K).to(torch.bfloat16)
    w = (rn(N, K) * 0.02).to(torch.bfloat16)
    b = rn(N)
    o = torch.empty(Mh, N, dtype=torch.bfloat16, device=dev)
    alias = a[:1].expand(Mh, K)
    for v in (5, 2):
        lib.visrep_set_gemm_variant(v)
        r = []
        for _ in range(3):
            r.append((t(lambda: engine.gemm(a, w, b, epi, act=act, out=o)), t(lambda: engine.gemm(alias, w, b, epi, act=act, out=o))))
        real = sorted(x[0] for x in r)[1]
        al = sorted(x[1] for x in r)[1]
        fl = 2.0 * Mh * N * K
        print(f"{name} v{v}: real A {real:.4f} ms ({fl / real / 1e9:7.1f} TF) | A rows aliased (lda = 0) {al:.4f} ms ({fl / al / 1e9:7.1f} TF)", flush=True)
    del a, w, o
lib.visrep_set_gemm_variant(5)
