"""CLIP image embeddings on MI355X: HF `CLIPVisionModelWithProjection(pixels).image_embeds` - what the image-variation
pipeline's `_encode_image` feeds the UNet as cross-attention context (vendored diffusers
pipeline_stable_diffusion_image_variation.py:133-141, called from dift_imsd.py:217-220).

    image_embeds = visual_projection(post_layernorm(last_hidden_state[:, 0]))

The encoder is the ViT tower engine (all layers this time); the head is one LayerNorm over the CLS rows and one GEMM.
`resize_bilinear` is F.interpolate(size=(224, 224), mode="bilinear") as the featurizer applies it (dift_imsd.py:215).
"""
from __future__ import annotations

import torch

from . import _lib
from .engine import VitEngine, gemm, layernorm
from .vit_weights import ViTSpec


def resize_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    """x [B, C, H, W] fp32|bf16 on the GPU -> [B, C, OH, OW] bf16."""
    lib = _lib.require_gpu()
    x = x.contiguous()
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    B, C, H, W = x.shape
    y = torch.empty(B, C, size[0], size[1], dtype=torch.bfloat16, device=x.device)
    rc = lib.visrep_resize_bilinear(_lib.ptr(x), _lib.F32 if x.dtype == torch.float32 else _lib.BF16, _lib.ptr(y), B * C, H, W, size[0],
                                    size[1], _lib.stream_ptr())
    _lib.check(rc, "visrep_resize_bilinear")
    return y


class ClipImageEmbedder:
    def __init__(self, spec: ViTSpec, weights: dict, post_ln_g, post_ln_b, projection, device=None):
        """weights: packed ViT weights (vit_weights.pack_hf_state_dict); projection [proj_dim, d] (no bias)."""
        self.spec = spec
        self.engine = VitEngine(spec, weights, device)
        dev = self.engine.device
        self.post = (post_ln_g.detach().to(dev, torch.float32).contiguous(), post_ln_b.detach().to(dev, torch.float32).contiguous())
        p = projection.detach().float()
        n = (p.shape[0] + 63) // 64 * 64
        w = torch.zeros(n, p.shape[1])
        w[: p.shape[0]] = p
        self.proj, self.proj_dim = w.to(dev, torch.bfloat16).contiguous(), p.shape[0]

    @torch.no_grad()
    def forward(self, pixels: torch.Tensor) -> torch.Tensor:
        """pixels [B, 3, S, S] -> image_embeds [B, proj_dim] bf16."""
        hid = self.engine.forward(pixels)                                   # last_hidden_state [B, T, d]
        cls = hid[:, 0]                                                     # strided view of the CLS rows
        n = layernorm(cls, self.post[0], self.post[1], self.spec.eps)
        return gemm(n, self.proj)[:, : self.proj_dim]
