import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
dev = "cuda:0"
M, d, m = 256 * 577, 1024, 4096
x = torch.randn(M, d, device=dev).to(torch.bfloat16)
w1 = (torch.randn(m, d, device=dev) * 0.02).to(torch.bfloat16)
b1 = torch.randn(m, device=dev)
o1 = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, fn in {"fc1 bias only": lambda: engine.gemm(x, w1, b1, _lib.EPI_BIAS, out=o1),
                 "fc1 no bias": lambda: engine.gemm(x, w1, None, _lib.EPI_BIAS, out=o1),
                 "fc1 quick_gelu": lambda: engine.gemm(x, w1, b1, _lib.EPI_ACT, act="quick_gelu", out=o1),
                 "fc1 gelu_erf": lambda: engine.gemm(x, w1, b1, _lib.EPI_ACT, act="gelu", out=o1),
                 "fc1 gelu_tanh": lambda: engine.gemm(x, w1, b1, _lib.EPI_ACT, act="gelu_tanh", out=o1)}.items():
    ms = t(fn)
    print(f"{name:16s} {ms:.4f} ms  {2.0 * M * m * d / ms / 1e9:.1f} TFLOP/s")
