"""A-score oracle (CPU).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Restates /root/reference/A_score/compute.py:
  * normalize_feat           compute.py:12-15
  * per-image score          compute.py:54-72  (cosine_similarity → max(dim=1) → mean)
  * per-encoder aggregation  compute.py:75-81  (python-float means, (a336+a224)/2)

The reference materialises the [Nt, Nr, D] broadcast product; the oracle uses
the algebraically identical matmul of the re-normalised rows (F.cosine_similarity
divides each operand by max(||x||, 1e-8) before the dot product).
"""
from __future__ import annotations

import torch


def normalize_feat(feat: torch.Tensor, epsilon: float = 1e-10) -> torch.Tensor:
    # compute.py:12-15
    norms = torch.linalg.norm(feat, dim=-1, keepdim=True)
    return feat / (norms + epsilon)


def max_cos_mean(other: torch.Tensor, ref: torch.Tensor) -> float:
    """mean_t max_s cos(other[t], ref[s])  — compute.py:54-72 for one image / one reference."""
    o = normalize_feat(other.float())
    r = normalize_feat(ref.float())
    # F.cosine_similarity(dim=-1, eps=1e-8) semantics
    o = o / torch.linalg.norm(o, dim=-1, keepdim=True).clamp_min(1e-8)
    r = r / torch.linalg.norm(r, dim=-1, keepdim=True).clamp_min(1e-8)
    sim = o @ r.t()
    return sim.max(dim=1).values.mean().item()


def _bf(x: torch.Tensor) -> torch.Tensor:
    """round to bf16, keep fp32 storage"""
    return x.to(torch.bfloat16).float()


def max_cos_mean_reference_arithmetic(other: torch.Tensor, ref: torch.Tensor) -> float:
    """compute.py:12-15,54-72 as torch executes them ON bf16 TENSORS (what the reference's dumped features are): every op
    computes in fp32 and rounds its result to bf16.  Restated with explicit roundings on fp32 tensors, op by op:
      normalize_feat            torch.norm -> bf16; + epsilon -> bf16; feat / (..) -> bf16                        (compute.py:12-15)
      F.cosine_similarity       per operand x / max(|x|, eps) with |x| and the quotient in bf16; the broadcast PRODUCT is a bf16
                                tensor [Nt, Nr, D]; its sum over D accumulates in fp32 and is rounded once         (compute.py:64-65)
      .max(dim=1).values.mean() ATen's bf16 mean = fp32 sum -> fp32 divide -> one rounding                         (compute.py:68-72)
    Returns the python float `.item()` gives (a bf16-representable value).  Checked bit for bit against the reference script run on bf16
    tensors (tests/golden/ascore.npz, cases bf16_inputs / bf16_wide / bf16_self; torch version recorded in the fixture)."""
    def normalize(x):
        xf = x.float()
        n = _bf(torch.sqrt((xf * xf).sum(-1, keepdim=True)))
        n = _bf(n + 1e-10)
        return _bf(xf / n)

    def cos_operand(x):
        n = _bf(torch.sqrt((x * x).sum(-1, keepdim=True)))
        n = torch.clamp_min(n, _bf(torch.tensor(1e-8)))
        return _bf(x / n)
    o = cos_operand(normalize(other.to(torch.bfloat16)))
    r = cos_operand(normalize(ref.to(torch.bfloat16)))
    sim = torch.empty(o.shape[0], r.shape[0])
    for t0 in range(0, o.shape[0], 64):                                   # row blocks: the [Nt, Nr, D] product is 5 GB at 576 x 576 x 4096
        sim[t0:t0 + 64] = _bf(_bf(o[t0:t0 + 64, None, :] * r[None, :, :]).sum(-1))
    return _bf(sim.max(dim=1).values.mean()).item()


def a_score(others, refs336, refs224) -> float:
    """compute.py:48-81 for one encoder: lists of per-image [N, D] tensors."""
    s336 = [max_cos_mean(o, r) for o, r in zip(others, refs336)]
    s224 = [max_cos_mean(o, r) for o, r in zip(others, refs224)]
    a336 = sum(s336) / len(s336)
    a224 = sum(s224) / len(s224)
    return (a336 + a224) / 2, a336, a224
