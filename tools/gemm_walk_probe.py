"""Tile-walk experiment of the persistent 256x256 GEMM (round 6): column-group-major walks (visrep_set_gemm_walk) against the
default 4 x 8 windows, at the headline shapes.  Two modes:
  time            every walk code in one process, 20 launches each, the whole list twice (order effects visible), TFLOP/s per shape
  pmc <code>      three launches of fc1 and Q|K with one walk code - for a `rocprofv3 --pmc FETCH_SIZE` pass (one pass per code: the counter
                  summary is per kernel name)
VISREP_LIB selects a variant library (e.g. the -DV5_NT_X=1 build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine  # noqa: E402

dev = "cuda:0"
B, T, d, m = 256, 577, 1024, 4096
M = 576 * 256                     # the rows of the full tile rounds (the 256q kernel alone, no 128x128 tail launches)
mode = sys.argv[1] if len(sys.argv) > 1 else "time"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16)
hm = torch.randn(M, m, device=dev, generator=g).to(torch.bfloat16)
w1 = (torch.randn(m, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
w2 = (torch.randn(d, m, device=dev, generator=g) * 0.02).to(torch.bfloat16)
wqk = (torch.randn(2 * d, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
wo = (torch.randn(d, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
b1 = torch.randn(m, device=dev, generator=g)
o1 = torch.empty(M, m, dtype=torch.bfloat16, device=dev)
o2 = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
oqk = torch.empty(M, 2 * d, dtype=torch.bfloat16, device=dev)
lib = _lib.load()
shapes = {
    "fc1": (lambda: engine.gemm(x, w1, b1, _lib.EPI_ACT, act="quick_gelu", out=o1), 2.0 * M * m * d),
    "qk": (lambda: engine.gemm(x, wqk, None, _lib.EPI_BIAS, out=oqk), 2.0 * M * 2 * d * d),
    "out": (lambda: engine.gemm(x, wo, None, _lib.EPI_RESID, resid=o2, out=o2), 2.0 * M * d * d),
    "fc2": (lambda: engine.gemm(hm, w2, None, _lib.EPI_RESID, resid=o2, out=o2), 2.0 * M * m * d),
}
if mode == "pmc":
    code = int(sys.argv[2])
    lib.visrep_set_gemm_walk(code)
    for _ in range(3):
        shapes["fc1"][0]()
        shapes["qk"][0]()
    torch.cuda.synchronize()
    print("walk", code, "lib", os.environ.get("VISREP_LIB", "default"))
    sys.exit(0)


def t(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


codes = [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,2,4,8,16".split(","))]
ref = None
for rnd in range(2):
    for code in codes:
        lib.visrep_set_gemm_walk(code)
        row = {}
        for name, (fn, fl) in shapes.items():
            ms = t(fn)
            row[name] = (round(ms, 4), round(fl / ms / 1e9, 1))
        if code == 0 and rnd == 0:
            o1_ref, oqk_ref = o1.clone(), oqk.clone()
        else:
            assert torch.equal(o1, o1_ref) and torch.equal(oqk, oqk_ref), "the walk changed the result"
        print(f"round {rnd} walk {code:3d} lib {os.path.basename(os.environ.get('VISREP_LIB', 'default'))}: " + "  ".join(f"{k} {v[0]} ms {v[1]} TF" for k, v in row.items()), flush=True)
lib.visrep_set_gemm_walk(0)
