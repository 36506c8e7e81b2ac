"""mm_projector oracle (CPU).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Restates /root/reference/llava/model/multimodal_projector/builder.py:40-47
(`mlp{N}x_gelu`: Linear → [GELU(erf) → Linear]*(N-1)) and :37-38 (`linear`).
"""
import torch
import torch.nn.functional as F


def mlp_gelu(x: torch.Tensor, weights, biases) -> torch.Tensor:
    h = x.float()
    for i, (w, b) in enumerate(zip(weights, biases)):
        if i > 0:
            h = F.gelu(h)  # nn.GELU() default = exact erf
        h = F.linear(h, w.float(), None if b is None else b.float())
    return h
