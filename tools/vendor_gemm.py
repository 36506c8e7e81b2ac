"""Context measurement: the vendor library (hipBLASLt behind torch.nn.functional.linear) on the ViT-L GEMM shapes of the bench, same box, same
HIP-event protocol as tools/gemm_v4_time.py.  Not used by the package; run under `rocprofv3 --kernel-trace --stats` to see which kernels it picks."""
import sys
import torch
import torch.nn.functional as F

M = 256 * 577 - 256 * 577 % 256 if len(sys.argv) < 2 else int(sys.argv[1])
SHAPES = [("fc1", 4096, 1024), ("qk", 2048, 1024), ("v/out", 1024, 1024), ("fc2", 1024, 4096)]
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, N, K in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.03
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        y = F.linear(x, w, b)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = F.linear(x, w, b)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 10)
    ms = sorted(best)[1]
    print(f"{name:6s} M={M} N={N} K={K}  {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s (F.linear + bias)", flush=True)
    del x, w, b, y
