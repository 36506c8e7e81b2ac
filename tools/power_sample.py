"""Board telemetry under the headline kernels: is the board at its power cap under the GEMMs?

Phases, each ~2.5 s of back-to-back launches with the sysfs sampler on (law_of_vision_representation_in_mllms_amd/telemetry.py: hwmon power / shader clock at
~100 Hz + the firmware's energy accumulator and PPT / thermal throttler residencies over the window):
  idle | mfma probe (free-running 16x16x32 stream, no memory) | fc1 (LayerNorm-folded, QuickGELU) | fc2 | Q|K | out-proj | attention |
  the whole forward (batch 256) | the vendor library (hipBLASLt behind F.linear) on the fc1 shape, for context.
Prints one JSON object; tools/gpu.sh py:tools/power_sample.py writes it under gpurun_out/.  profiles/round6_power.md is made from it."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine, telemetry  # noqa: E402
from law_of_vision_representation_in_mllms_amd import vit_weights as VW  # noqa: E402

SECONDS = float(os.environ.get("POWER_SECONDS", "2.5"))


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.load()
    sp = _lib.stream_ptr
    board = telemetry.Board(0)
    out = {"card": board.card, "selfcheck": board.selfcheck() if board.ok else None, "seconds_per_phase": SECONDS}
    spec = VW.SPECS["openai/clip-vit-large-patch14-336"]
    os.environ["VISREP_FAST_SYNTHETIC"] = "1"
    w = VW.synthetic_weights(spec, seed=1, n_layers=23)
    os.environ.pop("VISREP_FAST_SYNTHETIC")
    eng = engine.VitEngine(spec, w, dev)
    B = 256
    px = torch.randn(B, 3, 336, 336, device=dev).to(torch.bfloat16)
    M, d, m = B * spec.tokens, spec.d, spec.mlp
    x = eng.forward(px, n_layers=11).reshape(M, d).clone()
    L0 = w["layers"][11]
    w1 = (L0["w1"].float() * L0["ln2_g"].float()[None]).to(dev).to(torch.bfloat16)
    wqk = (L0["wqkv"].float() * L0["ln1_g"].float()[None])[: 2 * d].to(dev).to(torch.bfloat16).contiguous()
    w2, wo, b1 = L0["w2"].to(dev).to(torch.bfloat16), L0["wo"].to(dev).to(torch.bfloat16), L0["b1"].to(dev)
    o1 = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
    hm = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
    o2 = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
    oqk = torch.empty(M, 2 * d, dtype=torch.bfloat16, device=dev)
    rt = torch.zeros((M + 127) // 128 * 128 + 8, 2, dtype=torch.float32, device=dev)
    part = torch.empty(M, d // 64, 2, dtype=torch.float32, device=dev)
    _lib.check(lib.visrep_layernorm_stats(_lib.ptr(x), d, _lib.ptr(rt), M, d, 1e-5, sp()), "stats")
    s1, sqk = w1.float().sum(1).contiguous(), wqk.float().sum(1).contiguous()
    bqk = torch.zeros(2 * d, dtype=torch.float32, device=dev)

    def fc1():
        _lib.check(lib.visrep_gemm_bf16_ln(_lib.ptr(x), d, _lib.ptr(w1), d, _lib.ptr(b1), _lib.ptr(rt), _lib.ptr(s1), _lib.ptr(o1), m, M, m, d,
                                           _lib.EPI_ACT, _lib.ACT["quick_gelu"], sp()), "gemm_ln")

    def qk():
        _lib.check(lib.visrep_gemm_bf16_ln(_lib.ptr(x), d, _lib.ptr(wqk), d, _lib.ptr(bqk), _lib.ptr(rt), _lib.ptr(sqk), _lib.ptr(oqk), 2 * d, M, 2 * d, d,
                                           _lib.EPI_BIAS, 0, sp()), "gemm_ln")

    def resid(a, wt, K):
        return lambda: _lib.check(lib.visrep_gemm_bf16_resid_stats(_lib.ptr(a), K, _lib.ptr(wt), K, None, _lib.ptr(o2), d, M, d, K, _lib.ptr(o2), None,
                                                                   _lib.ptr(rt), _lib.ptr(part), 1e-5, sp()), "resid")
    fc1()
    hm.copy_(o1)
    fc2, outp = resid(hm, w2, m), resid(x, wo, d)
    qk()
    T = spec.tokens
    qa = oqk.clone()
    wv = L0["wqkv"][2 * d:].to(dev).to(torch.bfloat16).contiguous()
    vt = engine.gemm_rows(x, T - 1, T, 1, B * (T - 1), wv, None, epilogue=_lib.EPI_VT)
    vcls = engine.gemm_rows(x, 1, T, 0, B, wv, None)
    attn = lambda: engine.mhsa_cls(qa, vt, vcls, B, T, spec.heads)
    sink = torch.zeros(4, dtype=torch.int64, device=dev)
    flop = C.c_double(0.0)
    probe = lambda: _lib.check(lib.visrep_debug_mfma_probe(20000, 1, _lib.ptr(sink), C.byref(flop), sp()), "probe")
    outb = torch.empty(B, T, d, dtype=torch.bfloat16, device=dev)
    fwd = lambda: eng.forward(px, n_layers=23, out=outb)
    xr = torch.randn(M, d, device=dev).to(torch.bfloat16)
    wr = (torch.randn(m, d, device=dev) * 0.02).to(torch.bfloat16)
    vendor = lambda: torch.nn.functional.linear(xr, wr, out=None)
    fc1_n01 = lambda: engine.gemm(xr, wr, b1, _lib.EPI_ACT, act="quick_gelu", out=o1)

    def timed(fn, flops):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        return {"ms": round(ms, 4), "tflops": round(flops / (ms * 1e-3) / 1e12, 1)}

    # idle
    torch.cuda.synchronize()
    time.sleep(1.0)
    s = telemetry.Sampler(board).start()
    time.sleep(1.5)
    out["idle"] = s.stop()
    fl_img = 23 * (2 * T * d * 3 * d + 2 * T * d * d + 4 * T * T * d + 4 * T * d * m)
    phases = [("mfma_probe", probe, None), ("fc1", fc1, 2.0 * M * m * d), ("fc2", fc2, 2.0 * M * m * d), ("qk", qk, 2.0 * M * 2 * d * d),
              ("out_proj", outp, 2.0 * M * d * d), ("attention", attn, 4.0 * B * T * T * d), ("forward_b256", fwd, float(fl_img) * B),
              ("fc1_n01_operands", fc1_n01, 2.0 * M * m * d), ("vendor_fc1_shape", vendor, 2.0 * M * m * d)]
    for name, fn, fl in phases:
        ent = telemetry.measure(fn, SECONDS)
        if fl:
            ent["kernel"] = timed(fn, fl)
        elif name == "mfma_probe":
            probe()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            probe()
            e1.record()
            torch.cuda.synchronize()
            cyc, ticks = sink[1].item(), sink[2].item()
            ent["kernel"] = {"tflops": round(flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1), "clock_ghz_in_kernel": round(cyc / (ticks * 10.0), 3) if ticks else None}
        out[name] = ent
        time.sleep(0.5)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
