"""fc2-shape GEMM (N=1024, K=4096) throughput vs M (working-set size): is the kernel bound by HBM first-touch latency?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
dev = "cuda:0"
lib = _lib.load()
def t(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for N, K in ((1024, 4096), (4096, 1024)):
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    for M in (16384, 32768, 65536, 147712, 256 * 64 * 9):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        line = f"N={N} K={K} M={M:7d} (A {M*K*2/2**20:6.0f} MB, tiles/CU {M/256*N/256/256:5.2f})"
        for v in (1, 2, 3):
            lib.visrep_set_gemm_variant(v)
            ms = t(lambda: engine.gemm(a, w, None, _lib.EPI_BIAS, out=o))
            line += f"  v{v}: {ms:7.3f} ms {2.0*M*N*K/ms/1e9:7.1f} TF"
        print(line, flush=True)
        del a, o
