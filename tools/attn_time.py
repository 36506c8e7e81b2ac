"""Attention forward at the headline shape (256 images x 16 heads x 577 tokens, head width 64): the classic kernel (attn_fwd<1>, scale in the
exponent's FMA) and the pre-scaled-Q kernel the ViT towers launch (attn_fwd<1, PS>: scale folded into Q, reference in the accumulator init).
Other builds: VISREP_LIB=<path to a build.build_variant_lib() library> python tools/attn_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
B, T, H, d = 256, 577, 16, 1024
M = B * T
torch.manual_seed(0)
data = os.environ.get("ATTN_DATA", "randn")             # randn | zeros | peaked (q, k x 6: P mostly 0) | flat (q, k x 0.05: P ~ uniform) | vzero
qscale = {"peaked": 6.0, "flat": 0.05, "zeros": 0.0}.get(data, 1.0)
qk_f = torch.randn(M, 2 * d, device="cuda") * qscale
qk = qk_f.to(torch.bfloat16)
qk_f[:, :d] *= 0.125 * 1.4426950408889634              # what the folded Q projection writes
qk_ps = qk_f.to(torch.bfloat16)
del qk_f
x = (torch.randn(M, d, device="cuda") * (0.0 if data in ("zeros", "vzero") else 1.0)).to(torch.bfloat16)
w = (torch.randn(d, d, device="cuda") * 0.03).to(torch.bfloat16)
vt = engine.linear_vt(x, w, None)
tag = os.path.basename(os.environ.get("VISREP_LIB", "default"))
ref = None
vt_p = engine.gemm_rows(x, T - 1, T, 1, B * (T - 1), w, None, epilogue=_lib.EPI_VT)        # patch rows only, image-aligned columns
vcls = engine.gemm_rows(x, 1, T, 0, B, w, None)
for name, fn in (("classic", lambda: engine.mhsa(qk, vt, B, T, H, 0.125)), ("pre-scaled Q", lambda: engine.mhsa(qk_ps, vt, B, T, H, 0.0)),
                 ("image-aligned", lambda: engine.mhsa_cls(qk_ps, vt_p, vcls, B, T, H))):
    for _ in range(10): out = fn()
    torch.cuda.synchronize()
    if ref is None: ref = out.float()
    err = ((out.float() - ref).norm() / ref.norm()).item()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{tag} [{data}] {name}: {best:.4f} ms  {4.0 * B * T * T * d / best / 1e9:.1f} TFLOP/s  rel diff to the classic kernel {err:.2e}", flush=True)
