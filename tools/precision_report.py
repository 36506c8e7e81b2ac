"""How far does tower precision move the scores?  Full-size towers (random-init weights), the same images through
  (a) the fp32 CPU oracle, (b) the oracle run in bf16 (what the reference's bf16 towers do), (c) the fp32 HIP engine, (d) the bf16 HIP engine,
then the SAME score arithmetic on each feature set: A score (CLIP-L/14-224 tokens -> mlp2x_gelu -> vs CLIP-L/14-336 and CLIP-L/14-224
references) and C score (DINOv2-L @224 maps -> window soft-argmax transfer -> PCK hits).  Output: markdown for profiles/round2_precision.md."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import engine  # noqa: E402
from law_of_vision_representation_in_mllms_amd import vit_weights as VW  # noqa: E402
from oracle import ascore as OA, cscore as OC, projector as OP, vit as OV  # noqa: E402

torch.set_num_threads(min(32, os.cpu_count() or 1))
dev = "cuda:0"
os.environ["VISREP_FAST_SYNTHETIC"] = "1"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm()).item()


def tower(name, side, seed):
    base = VW.SPECS[name]
    native = base.at_resolution(base.pos_grid * base.patch) if base.pos_grid else base
    w0 = VW.synthetic_weights(native, seed=seed, n_layers=23)
    spec, w = VW.weights_at_resolution(native, w0, side)
    g = torch.Generator().manual_seed(seed)
    px = torch.randn(N, 3, side, side, generator=g).to(torch.bfloat16).float()
    f = {"oracle fp32": OV.tower_features(spec, w, px, 23, "patch"),
         "oracle bf16": OV.tower_features(spec, w, px, 23, "patch", dtype=torch.bfloat16).float(),
         "HIP fp32 split6": engine.VitEngineF32(spec, w, dev, products=6).forward(px.to(dev), n_layers=23)[:, 1:].float().cpu(),      # three planes, six products
         "HIP fp32 split4": engine.VitEngineF32(spec, w, dev, products=4).forward(px.to(dev), n_layers=23)[:, 1:].float().cpu(),      # two planes, four products
         "HIP fp32 split3": engine.VitEngineF32(spec, w, dev, products=3).forward(px.to(dev), n_layers=23)[:, 1:].float().cpu(),      # two planes, three products
         "HIP fp32 exact": engine.VitEngineF32(spec, w, dev, gemm="native").forward(px.to(dev), n_layers=23)[:, 1:].float().cpu(),    # exact-fp32 MFMA
         "HIP bf16": engine.VitEngine(spec, w, dev).forward(px.to(dev), n_layers=23)[:, 1:].float().cpu()}
    return f


print("# Tower precision vs scores (`python tools/precision_report.py`, full-size random-init towers, %d images).  `HIP fp32 splitN` = the "
      "reference-precision route with projections and attention as N split-bf16 plane-pair products (6: three planes, fp32-equivalent; 4 / 3: two "
      "planes = 16 significand bits), `HIP fp32 exact` = exact-fp32 MFMA throughout\n" % N)
clip224 = tower("openai/clip-vit-large-patch14", 224, 1)
clip336 = tower("openai/clip-vit-large-patch14-336", 336, 2)
dino = tower("facebook/dinov2-large", 224, 3)
VARS = ("oracle fp32", "oracle bf16", "HIP fp32 split6", "HIP fp32 split4", "HIP fp32 split3", "HIP fp32 exact", "HIP bf16")
print("| tower features vs the fp32 oracle (rel. L2) | " + " | ".join(VARS[1:]) + " |\n|---|" + "---|" * (len(VARS) - 1))
for nm, f in (("CLIP-L/14-224", clip224), ("CLIP-L/14-336", clip336), ("DINOv2-L @224", dino)):
    print(f"| {nm} | " + " | ".join(f"{rel(f[v], f['oracle fp32']):.2e}" for v in VARS[1:]) + " |")
# A score on each variant (projector fp32 on the CPU for all: isolates the tower's precision)
g = torch.Generator().manual_seed(7)
p0, p2 = torch.randn(4096, 1024, generator=g) * 0.03, torch.randn(4096, 4096, generator=g) * 0.015
b0, b2 = torch.randn(4096, generator=g) * 0.02, torch.randn(4096, generator=g) * 0.02
proj = lambda f: OP.mlp_gelu(f, [p0, p2], [b0, b2])
print("\n| A score (DINOv2-L tokens vs the CLIP336 / CLIP224 stacks, %d images) | value | rel. diff to oracle fp32 |\n|---|---|---|" % N)
base = None
for var in VARS:
    a = OA.a_score(list(proj(dino[var])), list(proj(clip336[var])), list(proj(clip224[var])))[0]
    base = a if base is None else base
    print(f"| {var} | {a:.6f} | {abs(a - base) / abs(base):.2e} |")
# C score: PCK hits of DINOv2-L maps over random pairs / key points.  Targets = the fp32 oracle's own predictions + U(-40, 40) px, so a
# realistic share of the key points sits near the thresholds (random features and random targets would give no hits at all).
rs = np.random.RandomState(5)
cases = []
NPAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for _ in range(NPAIRS):
    i, j = int(rs.randint(N)), int(rs.randint(N))
    k1 = torch.zeros(20, 3)
    k1[:, :2] = torch.from_numpy(rs.uniform(0, 839, (20, 2)).astype(np.float32))
    k1[:, 2] = 1
    cases.append((i, j, k1, float(rs.uniform(150, 700)), torch.from_numpy(rs.uniform(-40, 40, (20, 2)).astype(np.float32))))


def predict(var, i, j, k1):
    d1, d2 = OC.normalize_feats(dino[var][i][None]), OC.normalize_feats(dino[var][j][None])
    return OC.keypoint_transfer(d1, d2, OC.kpts_to_patch_idx(k1, 16), 16)


gts = []
for (i, j, k1, thr, noise) in cases:
    k2 = torch.ones(20, 3)
    k2[:, :2] = predict("oracle fp32", i, j, k1) + noise
    gts.append(k2)
print("\n| C score hits (DINOv2-L 16x16 maps, %d pairs x 20 key points" % NPAIRS + ", targets = the fp32 oracle's predictions + U(-40, 40) px) | hits @0.1 / 0.05 / 0.01 | key points | flips vs oracle fp32 | max prediction shift (px) |\n|---|---|---|---|---|")
ref_hits = None
for var in VARS:
    tot, nk, shift = np.zeros(3, np.int64), 0, 0.0
    per = []
    for (i, j, k1, thr, _), k2 in zip(cases, gts):
        xy = predict(var, i, j, k1)
        shift = max(shift, (xy - predict("oracle fp32", i, j, k1)).abs().max().item())
        _, nv, h = OC.pair_pck(xy, k1, k2, thr)
        per.append(h)
        tot += h.sum(dim=-1).numpy()
        nk += nv
    hmat = torch.cat(per, dim=1)
    ref_hits = hmat if ref_hits is None else ref_hits
    flips = (hmat != ref_hits).sum(dim=1).tolist()
    print(f"| {var} | {tot.tolist()} | {nk} | {flips} | {shift:.3f} |")
