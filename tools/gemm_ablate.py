"""Attribute GEMM v2 cycles: time the fc2 / fc1 shapes with parts of the kernel switched off (results invalid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
dev = "cuda:0"
M, d, m = 256 * 577, 1024, 4096
x = torch.randn(M, d, device=dev).to(torch.bfloat16)
hm = torch.randn(M, m, device=dev).to(torch.bfloat16)
w1 = (torch.randn(m, d, device=dev) * 0.02).to(torch.bfloat16)
w2 = (torch.randn(d, m, device=dev) * 0.02).to(torch.bfloat16)
o1 = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
o2 = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
lib = _lib.load()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lib.visrep_set_gemm_variant(variant)
print('GEMM variant', variant)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
names = {0: "full", 8: "full, no s_setprio", 16: "full, s_setprio on L segments", 1: "no MFMA", 2: "no LDS-DMA", 4: "no ds_read", 6: "MFMA + barriers only", 3: "ds_read + barriers only", 5: "LDS-DMA + barriers only", 7: "barriers only"}
for mask, name in names.items():
    lib.visrep_debug_gemm_ablation(mask)
    a = t(lambda: engine.gemm(hm, w2, None, _lib.EPI_BIAS, out=o2))
    b = t(lambda: engine.gemm(x, w1, None, _lib.EPI_BIAS, out=o1))
    print(f"mask {mask} {name:26s} fc2-shape {a:7.3f} ms   fc1-shape {b:7.3f} ms", flush=True)
lib.visrep_debug_gemm_ablation(0)
