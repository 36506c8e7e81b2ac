"""Launch each hot kernel once or twice at the BASELINE shapes (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine

dev = "cuda:0"
B, T, d, m, H = 256, 577, 1024, 4096, 16
M = B * T
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
x = (torch.randn(M, d, device=dev)).to(torch.bfloat16)
hm = (torch.randn(M, m, device=dev)).to(torch.bfloat16)
w1 = (torch.randn(m, d, device=dev) * 0.02).to(torch.bfloat16)
w2 = (torch.randn(d, m, device=dev) * 0.02).to(torch.bfloat16)
wqk = (torch.randn(2 * d, d, device=dev) * 0.02).to(torch.bfloat16)
wo = (torch.randn(d, d, device=dev) * 0.02).to(torch.bfloat16)
b1 = torch.randn(m, device=dev)
o1 = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
o2 = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
oqk = torch.empty(M, 2 * d, dtype=torch.bfloat16, device=dev)
lib = _lib.load()
for v in variants:
    lib.visrep_set_gemm_variant(v)
    for _ in range(reps):
        engine.gemm(x, w1, b1, _lib.EPI_ACT, act="quick_gelu", out=o1)          # fc1
        engine.gemm(hm, w2, None, _lib.EPI_RESID, resid=o2, out=o2)            # fc2
        engine.gemm(x, wqk, None, _lib.EPI_BIAS, out=oqk)                      # qk
        engine.gemm(x, wo, None, _lib.EPI_RESID, resid=o2, out=o2)             # out
        vt = engine.linear_vt(x, wo, None)                                     # v
qk_act = torch.randn(M, 2 * d, device=dev).to(torch.bfloat16)
for _ in range(reps):
    engine.mhsa(qk_act, vt, B, T, H, 0.125)
    engine.layernorm(x, b1[:d], b1[:d], 1e-5)
torch.cuda.synchronize()
print("probe done")
