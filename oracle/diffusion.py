"""CPU oracle of the Stable-Diffusion feature tower - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

Restates, in plain functional torch (NCHW, fp32 unless `dtype` says otherwise), what the reference executes for one
`SDFeaturizer.forward` (llava/model/multimodal_encoder/diffLVLM/src/models/dift_sd.py:239-276):

  repeat_interleave(ensemble)                                  dift_sd.py:251
  vae.encode(img).latent_dist.sample() * scaling_factor        dift_sd.py:172   (vendored diffusers autoencoder_kl.py /
                                                                                 vae.py Encoder, DiagonalGaussianDistribution)
  scheduler.add_noise(latents, randn, t)                       dift_sd.py:175-176 (scheduling_ddim.py:471-495)
  MyUNet2DConditionModel.forward(..., up_ft_indices)           dift_sd.py:9-155  (down blocks, mid block, up blocks until
                                                                                  max(up_ft_indices), capture up_ft[i])
  view(B, ensemble, c, h, w).mean(1)                           dift_sd.py:274-275
  DiffVisionTower.forward: [B,c,h,w] -> [B, h*w, c]            diffusion_encoder.py:84-88

The two `randn` draws of the reference (posterior sample, DDIM noise) are EXPLICIT inputs here - the reference
draws them on the GPU generator and they cannot be reproduced.  Prompt embeddings (`encode_prompt`, the CLIP text encoder)
are an explicit input as well.  Pinned against the reference's own MyUNet2DConditionModel + the vendored AutoencoderKL /
DDIMScheduler with tiny configs: tests/golden/sd_tiny.npz (tests/golden/make_golden.py gen_sd).
"""
import math

import torch
import torch.nn.functional as F


def _gn(x, w, p, groups, eps, silu=False):
    y = F.group_norm(x, groups, w[f"{p}.weight"], w[f"{p}.bias"], eps)
    return F.silu(y) if silu else y


def _conv(x, w, p, stride=1, padding=1):
    return F.conv2d(x, w[f"{p}.weight"], w[f"{p}.bias"], stride=stride, padding=padding)


def _lin(x, w, p, bias=True):
    return F.linear(x, w[f"{p}.weight"], w.get(f"{p}.bias") if bias else None)


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    # embeddings.py:27-67 with flip_sin_to_cos=True, downscale_freq_shift=0 (the SD UNet config)
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def resnet_block(x, temb, w, p, groups, eps):
    # resnet.py ResnetBlock2D.forward (time_embedding_norm "default", output_scale_factor 1)
    h = _gn(x, w, f"{p}.norm1", groups, eps, silu=True)
    h = _conv(h, w, f"{p}.conv1")
    if temb is not None:
        h = h + _lin(F.silu(temb), w, f"{p}.time_emb_proj")[:, :, None, None]
    h = _gn(h, w, f"{p}.norm2", groups, eps, silu=True)
    h = _conv(h, w, f"{p}.conv2")
    if f"{p}.conv_shortcut.weight" in w:
        x = _conv(x, w, f"{p}.conv_shortcut", padding=0)
    return x + h


def attention(q, k, v, heads):
    # attention_processor.py AttnProcessor / SlicedAttnProcessor: softmax(q k^T / sqrt(dh)) v per head
    B, Tq, d = q.shape
    dh = d // heads
    sp = lambda t: t.view(B, -1, heads, dh).transpose(1, 2)
    s = (sp(q) @ sp(k).transpose(-1, -2)) * dh ** -0.5
    return (torch.softmax(s, dim=-1) @ sp(v)).transpose(1, 2).reshape(B, Tq, d)


def _basic_block(h, ctx, w, b, heads, C):
    # attention.py BasicTransformerBlock (layer_norm, self-attn, cross-attn, GEGLU feed-forward)
    ln = lambda t, n: F.layer_norm(t, (C,), w[f"{b}.{n}.weight"], w[f"{b}.{n}.bias"], 1e-5)
    n1 = ln(h, "norm1")
    a = attention(_lin(n1, w, f"{b}.attn1.to_q", False), _lin(n1, w, f"{b}.attn1.to_k", False), _lin(n1, w, f"{b}.attn1.to_v", False), heads)
    h = h + _lin(a, w, f"{b}.attn1.to_out.0")
    n2 = ln(h, "norm2")
    a = attention(_lin(n2, w, f"{b}.attn2.to_q", False), _lin(ctx, w, f"{b}.attn2.to_k", False), _lin(ctx, w, f"{b}.attn2.to_v", False), heads)
    h = h + _lin(a, w, f"{b}.attn2.to_out.0")
    n3 = ln(h, "norm3")
    val, gate = _lin(n3, w, f"{b}.ff.net.0.proj").chunk(2, dim=-1)          # activations.py GEGLU
    return h + _lin(val * F.gelu(gate), w, f"{b}.ff.net.2")


def transformer_2d(x, ctx, w, p, heads, groups, linear, depth=1):
    # transformer_2d.py continuous-input path + attention.py BasicTransformerBlock (layer_norm, GEGLU feed-forward)
    B, C, H, W = x.shape
    res = x
    h = _gn(x, w, f"{p}.norm", groups, 1e-6)
    if linear:
        h = _lin(h.permute(0, 2, 3, 1).reshape(B, H * W, C), w, f"{p}.proj_in")
    else:
        h = _conv(h, w, f"{p}.proj_in", padding=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
    for k in range(depth):                                                 # transformer_layers_per_block (SDXL: 1 / 2 / 10)
        h = _basic_block(h, ctx, w, f"{p}.transformer_blocks.{k}", heads, C)
    if linear:
        h = _lin(h, w, f"{p}.proj_out").reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = _conv(h.reshape(B, H, W, C).permute(0, 3, 1, 2), w, f"{p}.proj_out", padding=0)
    return h + res


def unet_up_features(u, w, sample, t, ctx, up_ft_indices=(0,)):
    """MyUNet2DConditionModel.forward (dift_sd.py:9-155): returns {i: up_ft[i]} ([B, c, h, w])."""
    B = sample.shape[0]
    temb = timestep_embedding(torch.as_tensor(t).reshape(1).expand(B), u.block_out[0]).to(sample.dtype)
    temb = _lin(F.silu(_lin(temb, w, "time_embedding.linear_1")), w, "time_embedding.linear_2")
    g, eps = u.groups, u.eps
    h = _conv(sample, w, "conv_in")
    skips = [h]
    for i in range(len(u.block_out)):
        for j in range(u.layers_per_block):
            h = resnet_block(h, temb, w, f"down_blocks.{i}.resnets.{j}", g, eps)
            if u.down_types[i].startswith("CrossAttn"):
                h = transformer_2d(h, ctx, w, f"down_blocks.{i}.attentions.{j}", u.heads[i], g, u.linear_projection, u.depth(i))
            skips.append(h)
        if i != len(u.block_out) - 1:
            h = _conv(h, w, f"down_blocks.{i}.downsamplers.0.conv", stride=2)
            skips.append(h)
    h = resnet_block(h, temb, w, "mid_block.resnets.0", g, eps)
    h = transformer_2d(h, ctx, w, "mid_block.attentions.0", u.heads[-1], g, u.linear_projection, u.depth(len(u.block_out) - 1))
    h = resnet_block(h, temb, w, "mid_block.resnets.1", g, eps)
    out = {}
    rev_heads = tuple(reversed(u.heads))
    for i in range(max(up_ft_indices) + 1):
        L = u.layers_per_block + 1
        for j in range(L):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(h, temb, w, f"up_blocks.{i}.resnets.{j}", g, eps)
            if u.up_types[i].startswith("CrossAttn"):
                h = transformer_2d(h, ctx, w, f"up_blocks.{i}.attentions.{j}", rev_heads[i], g, u.linear_projection, u.depth(len(u.block_out) - 1 - i))
        if i != len(u.block_out) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")       # upsampling.py Upsample2D
            h = _conv(h, w, f"up_blocks.{i}.upsamplers.0.conv")
        if i in up_ft_indices:
            out[i] = h
    return out


def vae_encode_moments(v, w, img):
    """AutoencoderKL.encode -> (mean, logvar) of the posterior (vae.py Encoder + quant_conv)."""
    g = v.groups
    h = _conv(img, w, "encoder.conv_in")
    for i in range(len(v.block_out)):
        for j in range(v.layers_per_block):
            h = resnet_block(h, None, w, f"encoder.down_blocks.{i}.resnets.{j}", g, 1e-6)
        if i != len(v.block_out) - 1:
            h = F.pad(h, (0, 1, 0, 1))                                   # downsampling.py Downsample2D, padding=0
            h = _conv(h, w, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    h = resnet_block(h, None, w, "encoder.mid_block.resnets.0", g, 1e-6)
    a = "encoder.mid_block.attentions.0"                                  # attention_processor.py Attention, 1 head, residual
    B, C, H, W = h.shape
    n = _gn(h, w, f"{a}.group_norm", g, 1e-6).view(B, C, H * W).transpose(1, 2)
    o = attention(_lin(n, w, f"{a}.to_q"), _lin(n, w, f"{a}.to_k"), _lin(n, w, f"{a}.to_v"), 1)
    h = h + _lin(o, w, f"{a}.to_out.0").transpose(1, 2).reshape(B, C, H, W)
    h = resnet_block(h, None, w, "encoder.mid_block.resnets.1", g, 1e-6)
    h = _conv(_gn(h, w, "encoder.conv_norm_out", g, 1e-6, silu=True), w, "encoder.conv_out")
    if "quant_conv.weight" in w:                                          # SD3's VAE: use_quant_conv = False
        h = _conv(h, w, "quant_conv", padding=0)
    mean, logvar = h.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)                                # vae.py DiagonalGaussianDistribution


def noisy_latents(spec, mean, logvar, post_noise, ddim_noise, t):
    # latent_dist.sample() * scaling_factor ; DDIMScheduler.add_noise (scheduling_ddim.py:471-495)
    lat = (mean + torch.exp(0.5 * logvar) * post_noise) * spec.vae.scaling_factor
    ac = spec.sched.alphas_cumprod()[int(t)].to(lat.dtype)
    return ac ** 0.5 * lat + (1 - ac) ** 0.5 * ddim_noise


def sd_features(spec, w_unet, w_vae, img, prompt_embeds, post_noise, ddim_noise, t=1, up_ft_index=0, ensemble_size=1,
                dtype=torch.float32):
    """SDFeaturizer.forward + DiffVisionTower.forward: img [B,3,H,W] in [-1,1] -> [B, h*w, c].

    prompt_embeds [1, L, cross_dim]; post_noise / ddim_noise [B*ensemble, latent_c, H/8, W/8]."""
    cast = lambda d: {k: x.to(dtype) for k, x in d.items()}
    w_unet, w_vae = cast(w_unet), cast(w_vae)
    B = img.shape[0]
    x = img.repeat_interleave(ensemble_size, dim=0).to(dtype)
    mean, logvar = vae_encode_moments(spec.vae, w_vae, x)
    lat = noisy_latents(spec, mean, logvar, post_noise.to(dtype), ddim_noise.to(dtype), t)
    ctx = prompt_embeds.to(dtype).expand(B * ensemble_size, -1, -1)
    ft = unet_up_features(spec.unet, w_unet, lat, t, ctx, (up_ft_index,))[up_ft_index]
    _, c, h, wd = ft.shape
    ft = ft.view(B, ensemble_size, c, h, wd).mean(1)
    return ft.permute(0, 2, 3, 1).reshape(B, h * wd, c).float()


def imsd_features(spec, w_unet, w_vae, img, image_embeds, post_noise, ddim_noise, t=1, up_ft_index=0, ensemble_size=1,
                  dtype=torch.float32):
    """IMSDFeaturizer.forward (dift_imsd.py:199-229): as sd_features, but the cross-attention context is each image's own
    CLIP image embedding [B, 1, cross_dim] (computed by oracle/vit.py clip_image_embeds on the bilinearly resized image)."""
    cast = lambda d: {k: x.to(dtype) for k, x in d.items()}
    w_unet, w_vae = cast(w_unet), cast(w_vae)
    B = img.shape[0]
    x = img.repeat_interleave(ensemble_size, dim=0).to(dtype)
    mean, logvar = vae_encode_moments(spec.vae, w_vae, x)
    lat = noisy_latents(spec, mean, logvar, post_noise.to(dtype), ddim_noise.to(dtype), t)
    ctx = image_embeds.to(dtype).repeat_interleave(ensemble_size, dim=0)
    ft = unet_up_features(spec.unet, w_unet, lat, t, ctx, (up_ft_index,))[up_ft_index]
    _, c, h, wd = ft.shape
    ft = ft.view(B, ensemble_size, c, h, wd).mean(1)
    return ft.permute(0, 2, 3, 1).reshape(B, h * wd, c).float()
