"""Throughput of the A-score and C-score kernels at the SURVEY §8(d) sizes (BASELINE configs 3 and 4, one GPU's shard).

A score: per encoder [n, Nt, 4096] bf16 targets vs clip336 [n, 576, 4096] and clip224 [n, 256, 4096]; algorithmic work
2*Nt*(576+256)*4096 flop and (Nt+832)*8192 bytes per image.  C score: SPair-shaped synthetic set, DINOv2-L maps
[C=1024, P*P] fp32, 12,234 pairs over ~1,800 distinct images, K~U{3..20}; algorithmic bytes 2*P^2*C*4 per pair."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import ascore_ops, cscore_ops
dev = "cuda:0"

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

out = {}
n = int(os.environ.get("SCORE_BENCH_N", "1000"))
g = torch.Generator(device=dev).manual_seed(3)
r336 = torch.randn(n, 576, 4096, device=dev, generator=g).to(torch.bfloat16)
r224 = torch.randn(n, 256, 4096, device=dev, generator=g).to(torch.bfloat16)
s336, s224 = ascore_ops.row_scales(r336), ascore_ops.row_scales(r224)          # reference sets: normalised once for all encoders
for enc, Nt in (("CLIP-336", 576), ("SigLIP", 196), ("DINOv2-L", 256), ("SD1.5", 576)):
    o = torch.randn(n, Nt, 4096, device=dev, generator=g).to(torch.bfloat16)
    def both():                                                                # an encoder's tokens: normalised once for both references
        so = ascore_ops.row_scales(o)
        return ascore_ops.max_cos_mean(o, r336, so, s336), ascore_ops.max_cos_mean(o, r224, so, s224)
    sec = timed(both)
    fl = 2.0 * Nt * 832 * 4096 * n
    by = (Nt + 832) * 8192.0 * n
    out[f"A.{enc}"] = {"images_per_s": round(n / sec, 1), "TFLOP/s": round(fl / sec / 1e12, 1), "alg_GB/s": round(by / sec / 1e9, 1)}
    del o
del r336, r224
for P in (16, 24):
    C, n_img, n_pairs = 1024, 1800, 12234
    rs = np.random.RandomState(5)
    bank = torch.randn(n_img, C, P * P, device=dev, generator=g)
    i1 = torch.from_numpy(rs.randint(0, n_img, n_pairs).astype(np.int32))
    i2 = torch.from_numpy(rs.randint(0, n_img, n_pairs).astype(np.int32))
    nkp = torch.from_numpy(rs.randint(3, 21, n_pairs).astype(np.int32))
    idx = torch.from_numpy(rs.randint(0, P * P, (n_pairs, 20)).astype(np.int32))
    kps = torch.rand(n_pairs, 20, 3) * 839; kps[:, :, 2] = 1
    thr = torch.from_numpy(rs.uniform(150, 700, n_pairs))
    by = 2.0 * P * P * C * 4 * n_pairs
    for layout, bk, srt in (("cp", bank, True), ("pc", bank.transpose(1, 2).contiguous(), False), ("pc", None, True)):
        bk = bank.transpose(1, 2).contiguous() if bk is None else bk
        def run():
            xy = cscore_ops.transfer(bk, i1, i2, idx, nkp, P, layout=layout, sort_pairs=srt, packed=None if srt else False)
            return cscore_ops.pck_counts(xy, kps, kps, thr, nkp)
        sec = timed(run)
        out[f"C.P{P}.{layout}" + ("" if srt else ".dataset_order")] = {"pairs_per_s": round(n_pairs / sec, 1), "ms_all_pairs": round(sec * 1e3, 3), "alg_GB/s": round(by / sec / 1e9, 1),
                                   "frac_of_8TB/s": round(by / sec / 8e12, 4)}
    del bank, bk
print(json.dumps(out, indent=1))
