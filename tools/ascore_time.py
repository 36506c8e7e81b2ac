import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import ascore_ops
dev = "cuda:0"; n = 1000
g = torch.Generator(device=dev).manual_seed(3)
r336 = torch.randn(n, 576, 4096, device=dev, generator=g).to(torch.bfloat16)
r224 = torch.randn(n, 256, 4096, device=dev, generator=g).to(torch.bfloat16)
for Nt in (576, 196, 256):
    o = torch.randn(n, Nt, 4096, device=dev, generator=g).to(torch.bfloat16)
    f = lambda: (ascore_ops.max_cos_mean(o, r336), ascore_ops.max_cos_mean(o, r224))
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); sec = (time.perf_counter() - t0) / 5
    print(f"Nt={Nt}: {n/sec:9.0f} img/s  {2.0*Nt*832*4096*n/sec/1e12:6.1f} TFLOP/s  {(Nt+832)*8192.0*n/sec/1e9:7.1f} GB/s")
