"""Per-segment cycle accounting of GEMM v2 (diagnostic build): average cycles per K-tile spent in each segment by wave 0
(group 0) and wave 4 (group 1) of block 0.  Segments: L0, bar, M0, bar, L1, bar, M1(+epilogue), bar."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
dev = "cuda:0"
lib = _lib.load()
lib.visrep_set_gemm_variant(2)
buf = torch.zeros(16, dtype=torch.int64, device=dev)
lib.visrep_debug_gemm_timing_buffer(_lib.ptr(buf))
names = ["L0", "bar", "M0", "bar", "L1", "bar", "M1+epi", "bar"]
for (M, N, K, tag) in ((147456, 1024, 4096, "fc2-shape"), (147456, 4096, 1024, "fc1-shape")):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for mask in (0, 2, 1, 3):   # full | no LDS-DMA | no MFMA | reads + barriers only
        lib.visrep_debug_gemm_ablation(mask)
        for _ in range(2):
            buf.zero_()
            engine.gemm(a, w, None, _lib.EPI_BIAS, out=o)
            torch.cuda.synchronize()
        t = buf.cpu().tolist()
        ktiles = (M // 256) * (N // 256) // 256 * (K // 32)
        for g in range(2):
            seg = [t[g * 8 + i] / ktiles for i in range(8)]
            print(f"{tag} mask {mask} group {g}: " + "  ".join(f"{n}={v:6.0f}" for n, v in zip(names, seg)) + f"   total/K-tile={sum(seg):6.0f} cycles", flush=True)
lib.visrep_debug_gemm_ablation(0)
