"""C-score transfer at the SPair-71k size (12,234 pairs, DINOv2-L-shaped position-major bank [1800, P^2, 1024] fp32), HIP events.  With a
library built from tools/experiments/cscore_lds_keypoints_r4.patch (visrep_set_cscore_lds exported) it times the LDS-staged key-point variant
against the streaming form in the same process (round 4: 1.67 vs 1.38 ms at P = 16, 5.38 vs 3.32 at P = 24 - not kept)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, cscore_ops
dev = torch.device("cuda", 0)
lib = _lib.load()
for P in (16, 24):
    rs = np.random.RandomState(5)
    n_img, n = 1800, 12234
    bank = torch.randn(n_img, P * P, 1024, device=dev)
    i2 = np.repeat(np.arange(n_img), 7)[:n].astype(np.int32)             # every target serves ~7 pairs, grouped
    i1 = rs.randint(0, n_img, n).astype(np.int32)
    nkp = rs.randint(3, 21, n).astype(np.int32)
    idx = rs.randint(0, P * P, (n, 20)).astype(np.int32)
    t = [torch.from_numpy(a) for a in (i1, i2, idx, nkp)]
    packed = cscore_ops.packed_rows_on(dev, *t)
    ref = None
    has_knob = hasattr(lib, "visrep_set_cscore_lds")
    for mode in ((1, 0, 1, 0) if has_knob else (0, 0)):
        if has_knob: lib.visrep_set_cscore_lds(mode)
        fn = lambda: cscore_ops.transfer(bank, *t, P, layout="pc", packed=packed)
        for _ in range(3): out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = fn()
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = out
        print(f"P={P} lds={mode}: {e0.elapsed_time(e1) / 10:.3f} ms  same bits as the first run: {torch.equal(out, ref)}", flush=True)
    if has_knob: lib.visrep_set_cscore_lds(1)
