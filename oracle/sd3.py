"""CPU oracle of the SD3 (MMDiT) feature tower - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

Restates `SD3Featurizer.forward` (diffLVLM/src/models/dift_sd3.py:139-175), its pipeline body (:92-120) and
`MySD3Transformer2DModell.forward` (:10-91) over the vendored diffusers pieces it calls: embeddings.py PatchEmbed with
`pos_embed_max_size` (cropped sincos table), CombinedTimestepTextProjEmbeddings (:660-676), attention.py
JointTransformerBlock (MMDiT: adaLN-Zero on both streams, joint attention, per-stream GELU-tanh feed-forward; the last
block is `context_pre_only`), attention_processor.py JointAttnProcessor2_0, normalization.py AdaLayerNormZero /
AdaLayerNormContinuous, and the vendored FlowMatchEulerDiscreteScheduler.add_noise
(scheduling_flow_match_euler_discrete.py:192-210), which the reference calls with the RAW integer timestep:
    noisy = t * latents + (1 - t) * noise          (t = 261 -> 261 * latents - 260 * noise; restated as is)
Like the DiT featurizer, `forward` falls off its end without `return` (dift_sd3.py:175); the 2x2-unfolded block output is
returned here (6144 = 4 * 1536 channels, as `feature_hid_size_mapping` says).

Prompt embeddings [1, L, joint_dim] and pooled projections [1, pooled_dim] are explicit inputs (pipe.encode_prompt with
text_encoder_3 = None: CLIP-L | CLIP-G hidden_states[-2] padded to 4096 + 256 zero rows for the absent T5).
Pinned against the reference's MySD3Transformer2DModell with a tiny config: tests/golden/sd3_tiny.npz (gen_sd3).
"""
import torch
import torch.nn.functional as F

from . import diffusion as OD
from .dit import sincos_pos_embed, unfold_2x2


def cropped_pos_embed(c, gh, gw):
    # embeddings.py PatchEmbed.__init__ (grid = pos_embed_max_size, base_size = sample_size // patch) + cropped_pos_embed
    full = sincos_pos_embed(c.d, c.pos_max, c.pos_max, c.sample_size // c.patch).reshape(c.pos_max, c.pos_max, c.d)
    top, left = (c.pos_max - gh) // 2, (c.pos_max - gw) // 2
    return full[top: top + gh, left: left + gw].reshape(gh * gw, c.d)


def conditioning(c, w, t, pooled):
    # CombinedTimestepTextProjEmbeddings: Timesteps(256, flip_sin_to_cos, shift 0) -> MLP ; pooled -> Linear, SiLU, Linear ; sum
    lin = lambda v, n: F.linear(v, w[f"time_text_embed.{n}.weight"], w[f"time_text_embed.{n}.bias"])
    tp = OD.timestep_embedding(torch.full((pooled.shape[0],), int(t)), 256).to(pooled.dtype)
    te = lin(F.silu(lin(tp, "timestep_embedder.linear_1")), "timestep_embedder.linear_2")
    pe = lin(F.silu(lin(pooled, "text_embedder.linear_1")), "text_embedder.linear_2")
    return te + pe


def sd3_block_outputs(c, w, latents, t, prompt_embeds, pooled):
    """MySD3Transformer2DModell.forward: returns the sample-stream hidden states after every block ([B, N, D])."""
    B, _, H, W = latents.shape
    D, heads = c.d, c.heads
    gh, gw = H // c.patch, W // c.patch
    x = F.conv2d(latents, w["pos_embed.proj.weight"], w["pos_embed.proj.bias"], stride=c.patch).flatten(2).transpose(1, 2)
    x = (x + cropped_pos_embed(c, gh, gw)[None].to(x.dtype)).to(x.dtype)
    temb = conditioning(c, w, t, pooled.to(x.dtype).expand(B, -1))
    ctx = F.linear(prompt_embeds.to(x.dtype).expand(B, -1, -1), w["context_embedder.weight"], w["context_embedder.bias"])
    n_layers = min(c.layers, 1 + max(int(k.split(".")[1]) for k in w if k.startswith("transformer_blocks.")))
    ln = lambda v: F.layer_norm(v, (D,), None, None, 1e-6)
    outs = []
    for i in range(n_layers):
        p = f"transformer_blocks.{i}"
        lin = lambda v, n: F.linear(v, w[f"{p}.{n}.weight"], w[f"{p}.{n}.bias"])
        last = i == c.layers - 1
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = lin(F.silu(temb), "norm1.linear").chunk(6, dim=1)
        nx = ln(x) * (1 + sc_a[:, None]) + sh_a[:, None]
        if last:                                                          # AdaLayerNormContinuous: chunk -> scale, shift
            c_sc, c_sh = lin(F.silu(temb), "norm1_context.linear").chunk(2, dim=1)
            nc = ln(ctx) * (1 + c_sc)[:, None] + c_sh[:, None]
        else:
            c_sh_a, c_sc_a, c_g_a, c_sh_m, c_sc_m, c_g_m = lin(F.silu(temb), "norm1_context.linear").chunk(6, dim=1)
            nc = ln(ctx) * (1 + c_sc_a[:, None]) + c_sh_a[:, None]
        q = torch.cat([lin(nx, "attn.to_q"), lin(nc, "attn.add_q_proj")], dim=1)
        k = torch.cat([lin(nx, "attn.to_k"), lin(nc, "attn.add_k_proj")], dim=1)
        v = torch.cat([lin(nx, "attn.to_v"), lin(nc, "attn.add_v_proj")], dim=1)
        a = OD.attention(q, k, v, heads)
        N = x.shape[1]
        x = x + g_a[:, None] * lin(a[:, :N], "attn.to_out.0")
        n2 = ln(x) * (1 + sc_m[:, None]) + sh_m[:, None]
        x = x + g_m[:, None] * lin(F.gelu(lin(n2, "ff.net.0.proj"), approximate="tanh"), "ff.net.2")
        if not last:
            ctx = ctx + c_g_a[:, None] * lin(a[:, N:], "attn.to_add_out")
            nc2 = ln(ctx) * (1 + c_sc_m[:, None]) + c_sh_m[:, None]
            ctx = ctx + c_g_m[:, None] * lin(F.gelu(lin(nc2, "ff_context.net.0.proj"), approximate="tanh"), "ff_context.net.2")
        outs.append(x)
    return outs


def flow_noisy_latents(spec, mean, logvar, post_noise, noise, t):
    lat = (mean + torch.exp(0.5 * logvar) * post_noise) * spec.vae.scaling_factor        # dift_sd3.py:108
    return float(t) * lat + (1.0 - float(t)) * noise                                      # add_noise with raw t (:111)


def sd3_features(spec, w_core, w_vae, img, prompt_embeds, pooled, post_noise, noise, t=1, up_ft_index=-1, dtype=torch.float32):
    """SD3Featurizer.forward (+ the missing return) + DiffVisionTower.forward: img [B,3,H,W] -> [B, (h/2)(w/2), 4*D]."""
    cast = lambda d: {k: x.to(dtype) for k, x in d.items()}
    w_core, w_vae = cast(w_core), cast(w_vae)
    mean, logvar = OD.vae_encode_moments(spec.vae, w_vae, img.to(dtype))
    lat = flow_noisy_latents(spec, mean, logvar, post_noise.to(dtype), noise.to(dtype), t)
    ft = unfold_2x2(sd3_block_outputs(spec.core, w_core, lat, t, prompt_embeds, pooled)[up_ft_index])
    B, C, h, w = ft.shape
    return ft.permute(0, 2, 3, 1).reshape(B, h * w, C).float()
