"""Diagnostic: twenty image-aligned attention launches back to back (bench.py's kernels.mhsa.ms) with a fresh and with a reused output tensor."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
dev = "cuda:0"
B, T, H, d = 256, 577, 16, 1024
g = torch.Generator(device=dev).manual_seed(1)
qk = (torch.randn(B * T, 2 * d, device=dev, generator=g) * 0.5).to(torch.bfloat16)
x = (torch.randn(B * T, d, device=dev, generator=g) * 0.5).to(torch.bfloat16)
wv = (torch.randn(d, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
vt = engine.gemm_rows(x, T - 1, T, 1, B * (T - 1), wv, None, epilogue=_lib.EPI_VT)
vcls = engine.gemm_rows(x, 1, T, 0, B, wv, None)
lib = _lib.load()
out = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
def direct():
    _lib.check(lib.visrep_mhsa_cls_fwd(_lib.ptr(qk), qk.stride(0), _lib.ptr(vt), vt.stride(0), _lib.ptr(vcls), vcls.stride(0), _lib.ptr(out), out.stride(0), B, T, H, 64, _lib.stream_ptr()), "mhsa")
def timeit(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, th / reps * 1e3
for name, fn in (("engine.mhsa_cls (fresh output each call)", lambda: engine.mhsa_cls(qk, vt, vcls, B, T, H)), ("same launch into one output", direct),
                 ("torch.empty(302 MB) alone", lambda: torch.empty(B * T, d, dtype=torch.bfloat16, device=dev))):
    for rep in range(2):
        ms, host = timeit(fn)
        print(f"{name:45s} {ms:8.4f} ms per call on the device, {host:8.4f} ms of host time per call", flush=True)
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_reserved() >> 20, "MiB reserved")
