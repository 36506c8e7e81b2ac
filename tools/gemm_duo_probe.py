"""The duo kernel (gemm variant 6: two independent 4-wave workgroups per CU, 256 x 128 tiles, gemm_bf16_duo.hip) against the persistent 256x256
ping-pong kernel (variant 5) and the 128x128 kernel (variant 1: also two workgroups per CU) at the headline shapes - VERDICT r5 item 4.
Same process, the variants interleaved, twice; results of 6 checked against 5 (fp32 summation order differs: tolerance, not bits)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine  # noqa: E402

dev = "cuda:0"
d, m = 1024, 4096
M = int(sys.argv[1]) if len(sys.argv) > 1 else 576 * 256
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16)
hm = torch.randn(M, m, device=dev, generator=g).to(torch.bfloat16)
w1 = (torch.randn(m, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
w2 = (torch.randn(d, m, device=dev, generator=g) * 0.02).to(torch.bfloat16)
wqk = (torch.randn(2 * d, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
wo = (torch.randn(d, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
b1 = torch.randn(m, device=dev, generator=g)
r0 = torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16)
o1 = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
o2 = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
oqk = torch.empty(M, 2 * d, dtype=torch.bfloat16, device=dev)
lib = _lib.load()
shapes = {
    "fc1": (lambda: engine.gemm(x, w1, b1, _lib.EPI_ACT, act="quick_gelu", out=o1), 2.0 * M * m * d, o1),
    "qk": (lambda: engine.gemm(x, wqk, None, _lib.EPI_BIAS, out=oqk), 2.0 * M * 2 * d * d, oqk),
    "out": (lambda: engine.gemm(x, wo, None, _lib.EPI_RESID, resid=r0, out=o2), 2.0 * M * d * d, o2),
    "fc2": (lambda: engine.gemm(hm, w2, None, _lib.EPI_RESID, resid=r0, out=o2), 2.0 * M * m * d, o2),
}


def t(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ref = {}
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "5,6,1".split(","))]
for rnd in range(2):
    for v in variants:
        lib.visrep_set_gemm_variant(v)
        row = {}
        for name, (fn, fl, out) in shapes.items():
            ms = t(fn)
            row[name] = (round(ms, 4), round(fl / ms / 1e9, 1))
            if v == 5 and rnd == 0:
                ref[name] = out.float().clone()
            elif rnd == 0 and name in ref:
                err = ((out.float() - ref[name]).norm() / ref[name].norm()).item()
                assert err < 2e-3, (v, name, err)
                row[name] += (f"rel {err:.1e}",)
        print(f"round {rnd} variant {v}: " + "  ".join(f"{k} {v_[0]} ms {v_[1]} TF" + (f" ({v_[2]})" if len(v_) > 2 else "") for k, v_ in row.items()), flush=True)
lib.visrep_set_gemm_variant(5)
