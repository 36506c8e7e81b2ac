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


def a_score(others, refs336, refs224) -> float:
    """compute.py:48-81 for one encoder: lists of per-image [N, D] tensors."""
    s336 = [max_cos_mean(o, r) for o, r in zip(others, refs336)]
    s224 = [max_cos_mean(o, r) for o, r in zip(others, refs224)]
    a336 = sum(s336) / len(s336)
    a224 = sum(s224) / len(s224)
    return (a336 + a224) / 2, a336, a224
