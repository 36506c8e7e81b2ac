"""CPU oracle of the DiT feature tower - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

Restates `DiTFeaturizer.forward` (llava/model/multimodal_encoder/diffLVLM/src/models/dift_dit.py:168-196), its pipeline
body (:126-144), `MyDiTTransformer2DModel.forward` (:18-124) and the timestep-only conditioning override
`MyCombinedTimestepLabelEmbeddings` (:9-16), on top of the vendored diffusers blocks it calls: embeddings.py PatchEmbed +
get_2d_sincos_pos_embed (:70-128), normalization.py AdaLayerNormZero (:51-84), attention.py BasicTransformerBlock
(ada_norm_zero branch, :214-315), activations.py GELU(approximate="tanh").  The VAE / DDIM part is oracle/diffusion.py.

NOTE the reference's `DiTFeaturizer.forward` ends without a `return` (dift_dit.py:196), so `DiffVisionTower.forward` would
crash on `None.shape`; the evident intent - returning the 2x2-unfolded block output [B, 4*D, h/2, w/2] (4608 channels, as
`feature_hid_size_mapping` says) - is what is restated here.  "dit does not enable ensemble" (:196): ensemble_size = 1.

Pinned against the reference's MyDiTTransformer2DModel with a tiny config: tests/golden/dit_tiny.npz (make_golden.py gen_dit).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import diffusion as OD


def sincos_pos_embed(dim: int, grid_h: int, grid_w: int, base_size: int) -> torch.Tensor:
    # embeddings.py:70-128 (interpolation_scale 1); note the first half encodes the *w* grid (meshgrid "w goes first")
    gh = np.arange(grid_h, dtype=np.float32) / (grid_h / base_size)
    gw = np.arange(grid_w, dtype=np.float32) / (grid_w / base_size)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_w, grid_h])

    def one(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(dim // 2, grid[0]), one(dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float()


def timestep_proj(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    # embeddings.py:27-67 with flip_sin_to_cos=True, downscale_freq_shift=1 (CombinedTimestepLabelEmbeddings.time_proj)
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - 1)
    emb = t[:, None].float() * torch.exp(exponent)[None]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def dit_block_outputs(c, w, latents, t):
    """MyDiTTransformer2DModel.forward: latents [B, 4, h, w] -> list of hidden states after every block ([B, N, D])."""
    B, _, H, W = latents.shape
    D = c.d
    gh, gw = H // c.patch, W // c.patch
    x = F.conv2d(latents, w["pos_embed.proj.weight"], w["pos_embed.proj.bias"], stride=c.patch).flatten(2).transpose(1, 2)
    x = (x + sincos_pos_embed(D, gh, gw, c.sample_size // c.patch)[None].to(x.dtype)).to(x.dtype)
    tp = timestep_proj(torch.full((B,), int(t)), 256).to(x.dtype)
    outs = []
    n_layers = min(c.layers, 1 + max(int(k.split(".")[1]) for k in w if k.startswith("transformer_blocks.")))   # truncated weight sets
    for i in range(n_layers):
        p = f"transformer_blocks.{i}"
        lin = lambda v, n: F.linear(v, w[f"{p}.{n}.weight"], w[f"{p}.{n}.bias"])
        e = f"norm1.emb.timestep_embedder"
        cond = lin(F.silu(lin(tp, f"{e}.linear_1")), f"{e}.linear_2")                       # timestep only (dift_dit.py:9-16)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = lin(F.silu(cond), "norm1.linear").chunk(6, dim=1)
        n1 = F.layer_norm(x, (D,), None, None, 1e-6) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        a = OD.attention(lin(n1, "attn1.to_q"), lin(n1, "attn1.to_k"), lin(n1, "attn1.to_v"), c.heads)
        x = gate_msa[:, None] * lin(a, "attn1.to_out.0") + x
        n3 = F.layer_norm(x, (D,), None, None, c.eps) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        x = gate_mlp[:, None] * lin(F.gelu(lin(n3, "ff.net.0.proj"), approximate="tanh"), "ff.net.2") + x
        outs.append(x)
    return outs


def unfold_2x2(ft):
    """dift_dit.py:191-195: [B, N, D] -> [B, 4*D, h/2, w/2]."""
    B = ft.shape[0]
    h = w = int(ft.shape[-2] ** 0.5)
    ft = ft.transpose(2, 1).reshape(B, -1, h, w)
    ft = ft.unfold(3, 2, 2).unfold(2, 2, 2)
    return ft.reshape(B, -1, h // 2, w // 2, 4).permute(0, 4, 1, 2, 3).reshape(B, -1, h // 2, w // 2)


def dit_features(spec, w_dit, w_vae, img, post_noise, ddim_noise, t=1, up_ft_index=-1, dtype=torch.float32):
    """DiTFeaturizer.forward (+ the missing return) + DiffVisionTower.forward: img [B,3,H,W] -> [B, (h/2)(w/2), 4*D]."""
    cast = lambda d: {k: x.to(dtype) for k, x in d.items()}
    w_dit, w_vae = cast(w_dit), cast(w_vae)
    mean, logvar = OD.vae_encode_moments(spec.vae, w_vae, img.to(dtype))
    lat = OD.noisy_latents(spec, mean, logvar, post_noise.to(dtype), ddim_noise.to(dtype), t)
    ft = unfold_2x2(dit_block_outputs(spec.core, w_dit, lat, t)[up_ft_index])
    B, C, h, w = ft.shape
    return ft.permute(0, 2, 3, 1).reshape(B, h * w, C).float()
