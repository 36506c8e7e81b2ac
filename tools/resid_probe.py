import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
dev = "cuda:0"
M, d, m = 256 * 577, 1024, 4096
x = torch.randn(M, d, device=dev).to(torch.bfloat16)
hm = torch.randn(M, m, device=dev).to(torch.bfloat16)
wo = (torch.randn(d, d, device=dev) * 0.02).to(torch.bfloat16)
w2 = (torch.randn(d, m, device=dev) * 0.02).to(torch.bfloat16)
b = torch.randn(d, device=dev)
o = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
def t(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(2):
    a = t(lambda: engine.gemm(x, wo, b, _lib.EPI_RESID, resid=o, out=o))
    c = t(lambda: engine.gemm(hm, w2, b, _lib.EPI_RESID, resid=o, out=o))
    e = t(lambda: engine.gemm(x, wo, b, _lib.EPI_BIAS, out=o))
print(f"out RESID {a:.4f} ms ({2.0*M*d*d/a/1e9:.0f} TF)  fc2 RESID {c:.4f} ms ({2.0*M*d*m/c/1e9:.0f} TF)  out-shaped BIAS {e:.4f} ms")
