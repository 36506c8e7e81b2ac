"""Device entry point of the A score (visrep_ascore_maxcos).  See csrc/ascore.hip; reference A_score/compute.py:54-72."""
from __future__ import annotations

import torch

from . import _lib


@torch.no_grad()
def max_cos_mean(other: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """scores[i] = mean_t max_s cos(other[i, t], ref[i, s]);  other [n, Nt, D], ref [n, Nr, D] on the GPU.

    bf16 inputs run the bf16 MFMA path (products exact, fp32 accumulate); anything else is upcast to fp32 and
    runs the exact-fp32 MFMA path (the parity definition of SURVEY.md F4).
    """
    lib = _lib.require_gpu()
    if other.dim() != 3 or ref.dim() != 3 or other.shape[0] != ref.shape[0] or other.shape[2] != ref.shape[2]:
        raise ValueError("expected other [n, Nt, D] and ref [n, Nr, D]")
    if other.dtype == torch.bfloat16 and ref.dtype == torch.bfloat16 and other.shape[2] % 16 == 0:
        dt = _lib.BF16
    else:
        other, ref, dt = other.float(), ref.float(), _lib.F32
    other, ref = other.contiguous(), ref.contiguous()
    n, Nt, D = other.shape
    Nr = ref.shape[1]
    scores = torch.empty(n, dtype=torch.float32, device=other.device)
    ws = torch.empty(lib.visrep_ascore_workspace_bytes(n, Nt, Nr), dtype=torch.uint8, device=other.device)
    rc = lib.visrep_ascore_maxcos(_lib.ptr(other), _lib.ptr(ref), n, Nt, Nr, D, dt, _lib.ptr(scores), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "visrep_ascore_maxcos")
    return scores
