"""Device entry points of the A score (visrep_ascore_maxcos*).  See csrc/ascore.hip; reference A_score/compute.py:54-72."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _typed(x: torch.Tensor):
    """bf16 inputs run the bf16 MFMA path (products exact, fp32 accumulate); anything else is upcast to fp32 and runs the
    exact-fp32 MFMA path (the parity definition of SURVEY.md F4)."""
    if x.dtype == torch.bfloat16 and x.shape[-1] % 16 == 0:
        return x.contiguous(), _lib.BF16
    return x.float().contiguous(), _lib.F32


@torch.no_grad()
def row_scales(x: torch.Tensor) -> torch.Tensor:
    """The per-row factor 1/(|x|+1e-10) / max(|x|/(|x|+1e-10), 1e-8) of [n, N, D] tokens -> fp32 [n, N] (compute.py:12-15 + the
    eps clamp of F.cosine_similarity).  Pass it to max_cos_mean when a tensor is scored more than once - the clip336 / clip224
    reference sets against every encoder, an encoder's tokens against both references."""
    lib = _lib.require_gpu()
    if x.dim() != 3:
        raise ValueError("expected [n, N, D]")
    x, dt = _typed(x)
    out = torch.empty(x.shape[:2], dtype=torch.float32, device=x.device)
    _lib.check(lib.visrep_ascore_row_scale(_lib.ptr(x), x.shape[0] * x.shape[1], x.shape[2], dt, _lib.ptr(out), _lib.stream_ptr()), "visrep_ascore_row_scale")
    return out


@torch.no_grad()
def max_cos_mean(other: torch.Tensor, ref: torch.Tensor, other_scale: Optional[torch.Tensor] = None,
                 ref_scale: Optional[torch.Tensor] = None, arithmetic: str = "exact") -> torch.Tensor:
    """scores[i] = mean_t max_s cos(other[i, t], ref[i, s]);  other [n, Nt, D], ref [n, Nr, D] on the GPU.
    other_scale / ref_scale: row_scales() of the same tensors (same dtype path), or None to compute them in this call.
    arithmetic: "exact" (default) = exact products, fp32 accumulation (MFMA; the parity definition on fp32-upcast inputs);
    "reference" = the arithmetic torch runs A_score/compute.py:12-15,54-72 in when the dumped tensors are bf16 - every op's result rounded
    to bf16 (norm, divide, each product inside F.cosine_similarity, their sum, the mean), i.e. the numbers the reference PRINTS
    (policy/ablations_t.csv: 1.0078125 for CLIP336 against itself).  bf16 inputs only; the scores come back as fp32 holding bf16 values."""
    lib = _lib.require_gpu()
    if other.dim() != 3 or ref.dim() != 3 or other.shape[0] != ref.shape[0] or other.shape[2] != ref.shape[2]:
        raise ValueError("expected other [n, Nt, D] and ref [n, Nr, D]")
    if arithmetic not in ("exact", "reference"):
        raise ValueError(f"arithmetic must be 'exact' or 'reference', got {arithmetic!r}")
    if arithmetic == "reference":
        if other.dtype != torch.bfloat16 or ref.dtype != torch.bfloat16:
            raise ValueError("arithmetic='reference' reproduces torch's bf16 op chain: both tensors must be bf16 (fp32 tensors: the default "
                             "'exact' mode already is the reference's fp32 arithmetic up to summation order)")
        if other_scale is not None or ref_scale is not None:
            raise ValueError("arithmetic='reference' normalises in bf16 itself: precomputed fp32 row scales do not apply")
        other, ref = other.contiguous(), ref.contiguous()
        n, Nt, D = other.shape
        Nr = ref.shape[1]
        scores = torch.empty(n, dtype=torch.float32, device=other.device)
        ws = torch.empty(lib.visrep_ascore_refarith_workspace_bytes(n, Nt, Nr, D), dtype=torch.uint8, device=other.device)
        _lib.check(lib.visrep_ascore_maxcos_refarith(_lib.ptr(other), _lib.ptr(ref), n, Nt, Nr, D, _lib.ptr(scores), _lib.ptr(ws), _lib.stream_ptr()),
                   "visrep_ascore_maxcos_refarith")
        return scores
    if other.dtype == torch.bfloat16 and ref.dtype == torch.bfloat16 and other.shape[2] % 16 == 0:
        dt = _lib.BF16
    else:
        if (other_scale is not None and other.dtype == torch.bfloat16) or (ref_scale is not None and ref.dtype == torch.bfloat16):
            raise ValueError("precomputed scales of a bf16 tensor cannot be used on the fp32 path (mixed dtypes)")
        other, ref, dt = other.float(), ref.float(), _lib.F32
    other, ref = other.contiguous(), ref.contiguous()
    n, Nt, D = other.shape
    Nr = ref.shape[1]
    for sc, rows in ((other_scale, Nt), (ref_scale, Nr)):
        if sc is not None and (sc.dtype != torch.float32 or tuple(sc.shape) != (n, rows) or not sc.is_contiguous() or sc.device != other.device):
            raise ValueError("scale must be the contiguous fp32 [n, N] tensor row_scales() returned for this tensor")
    scores = torch.empty(n, dtype=torch.float32, device=other.device)
    ws = torch.empty(lib.visrep_ascore_workspace_bytes(n, Nt, Nr), dtype=torch.uint8, device=other.device)
    rc = lib.visrep_ascore_maxcos_scaled(_lib.ptr(other), _lib.ptr(ref), _lib.ptr(other_scale), _lib.ptr(ref_scale), n, Nt, Nr, D, dt,
                                         _lib.ptr(scores), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "visrep_ascore_maxcos_scaled")
    return scores


@torch.no_grad()
def max_cos_mean_cpu(other: torch.Tensor, ref: torch.Tensor, threads: int = 0) -> torch.Tensor:
    """The A score's per-image term on HOST cores (visrep_ascore_maxcos_cpu: plain C++ fp32, csrc/host_twins.hip) - the `*_cpu` twin of
    SURVEY §8b for boxes without a GPU.  Explicit entry point only: max_cos_mean() never falls back to it.  CPU tensors in (any float
    dtype, upcast to fp32 = the parity definition of SURVEY F4), fp32 [n] out."""
    lib = _lib.load()
    if other.dim() != 3 or ref.dim() != 3 or other.shape[0] != ref.shape[0] or other.shape[2] != ref.shape[2]:
        raise ValueError("expected other [n, Nt, D] and ref [n, Nr, D]")
    if other.device.type != "cpu" or ref.device.type != "cpu":
        raise ValueError("max_cos_mean_cpu takes CPU tensors (the device path is max_cos_mean)")
    o, r = other.float().contiguous(), ref.float().contiguous()
    n, Nt, D = o.shape
    scores = torch.empty(n, dtype=torch.float32)
    _lib.check(lib.visrep_ascore_maxcos_cpu(_lib.ptr(o), _lib.ptr(r), n, Nt, r.shape[1], D, _lib.ptr(scores), int(threads)), "visrep_ascore_maxcos_cpu")
    return scores
