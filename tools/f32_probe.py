"""fp32 (exact-fp32 MFMA) path at the ViT-L/14-336 shapes: the four GEMMs of a layer alone, and the whole fp32 tower per stage
(rocprofv3 --kernel-trace --stats -- python tools/f32_probe.py lists the kernels).  Usage: python tools/f32_probe.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
from law_of_vision_representation_in_mllms_amd import vit_weights as VW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, d, m = 577, 1024, 4096
M = B * T
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, d, device=dev, generator=g)
h = torch.randn(M, m, device=dev, generator=g)
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for name, a, N, K, epi, act in (("fc1", x, m, d, _lib.EPI_ACT, "quick_gelu"), ("fc2", h, d, m, _lib.EPI_BIAS, "none"), ("qkv", x, 3 * d, d, _lib.EPI_BIAS, "none"),
                                ("out", x, d, d, _lib.EPI_BIAS, "none")):
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    b = torch.randn(N, device=dev, generator=g)
    o = torch.empty(M, N, device=dev)
    sec = ev(lambda: engine.gemm_f32(a, w, b, epi, act, out=o))
    print(f"gemm_f32 {name}: M={M} N={N} K={K}: {sec * 1e3:.3f} ms  {2.0 * M * N * K / sec / 1e12:.1f} TFLOP/s  ({2.0 * M * N * K / sec / 157.3e12:.3f} of the fp32 MFMA roof)", flush=True)
    # the same product on the bf16 matrix pipe: three planes / six plane-pair products (6x the bf16 FLOP), two planes / four and three products
    act_s = act if epi == _lib.EPI_ACT else "none"
    ref = engine.gemm_f32(a[:4096], w, b, epi, act)
    for products in (6, 4, 3):
        npl = engine.split_planes(products)
        ap, wp = engine.split_bf16_planes(a, npl), engine.split_bf16_planes(w, npl)
        sec_sp = ev(lambda: engine.split_bf16_planes(a, npl))
        sec = ev(lambda: engine.gemm_f32_split(ap, wp, b, act=act_s, out=o, products=products))
        got = engine.gemm_f32_split(ap[:4096], wp, b, act=act_s, products=products)
        print(f"   split-bf16 x{products} {name}: {sec * 1e3:.3f} ms  {2.0 * M * N * K / sec / 1e12:.1f} fp32-equivalent TFLOP/s = {2.0 * products * M * N * K / sec / 1e12:.0f} bf16 TFLOP/s "
              f"({2.0 * products * M * N * K / sec / 2500e12:.3f} of the bf16 roof); split pass of A alone {sec_sp * 1e3:.3f} ms; |split - exact| / |exact| = "
              f"{((got - ref).double().norm() / ref.double().norm()).item():.2e}", flush=True)
        del ap, wp
spec = VW.SPECS["openai/clip-vit-large-patch14-336"]
wts = VW.synthetic_weights(spec, seed=1, n_layers=23)
px = torch.randn(B, 3, 336, 336, device=dev, generator=g)
fl = 23 * (2 * T * d * 3 * d + 2 * T * d * d + 4 * T * T * d + 4 * T * d * m) * B
outs = {}
for route in ("native", "split6", "split4", "split3"):
    eng = engine.VitEngineF32(spec, wts, dev, gemm=route[:5] if route != "native" else route, products=int(route[5:]) if route != "native" else None)
    sec = ev(lambda: eng.forward(px, n_layers=23), reps=2)
    outs[route] = eng.forward(px[:4], n_layers=23)
    print(f"fp32 tower ({route} GEMMs) batch {B}: {sec * 1e3:.1f} ms = {B / sec:.1f} images/s = {fl / sec / 1e12:.1f} fp32-equivalent TFLOP/s "
          f"({fl / sec / 157.3e12:.3f} of the exact-fp32 MFMA roof)", flush=True)
    del eng
for r in ("split6", "split4", "split3"):
    print(f"{r} vs native features: rel L2 {((outs[r] - outs['native']).double().norm() / outs['native'].double().norm()).item():.2e}")
