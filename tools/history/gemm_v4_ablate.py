"""Where do v4's cycles go: time the K = 4096 and K = 1024 shapes with parts of the stream compiled out (results invalid).
Build the libraries on the build host first: python tools/gemm_v4_ablate.py build ; then on the GPU: python tools/gemm_v4_ablate.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MASKS = {0: "full", 1: "no LDS-DMA", 2: "no fragment reads", 3: "MFMA + barrier only", 4: "no MFMA", 6: "LDS-DMA + barrier only", 5: "reads + barrier only"}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from law_of_vision_representation_in_mllms_amd import build
    for m in MASKS:
        print(build.build_variant_lib(f"v4abl{m}", [f"-DV4_ABL={m}"]))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import torch
    from law_of_vision_representation_in_mllms_amd import _lib, engine
    dev = "cuda:0"
    Mh, d, m = 147456, 1024, 4096
    x = torch.randn(Mh, d, device=dev).to(torch.bfloat16)
    hm = torch.randn(Mh, m, device=dev).to(torch.bfloat16)
    w2 = (torch.randn(d, m, device=dev) * 0.02).to(torch.bfloat16)
    wo = (torch.randn(d, d, device=dev) * 0.02).to(torch.bfloat16)
    w1 = (torch.randn(m, d, device=dev) * 0.02).to(torch.bfloat16)
    of = torch.empty(Mh, d, dtype=torch.float32, device=dev)
    o1 = torch.empty(Mh, m, dtype=torch.float32, device=dev)
    _lib.load().visrep_set_gemm_variant(4)

    def t(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    a = t(lambda: engine.gemm(hm, w2, None, _lib.EPI_F32, out=of))
    b = t(lambda: engine.gemm(x, wo, None, _lib.EPI_F32, out=of))
    c = t(lambda: engine.gemm(x, w1, None, _lib.EPI_F32, out=o1))
    print(f"{sys.argv[2]:24s} Mx1024x4096 {a:7.4f} ms | Mx1024x1024 {b:7.4f} ms | Mx4096x1024 {c:7.4f} ms", flush=True)
    sys.exit(0)
for m, name in MASKS.items():
    lib = os.path.join(ROOT, "law_of_vision_representation_in_mllms_amd", f"libvisrep_hip_v4abl{m}.so")
    subprocess.run([sys.executable, __file__, "one", name], env=dict(os.environ, VISREP_LIB=lib), check=False)
