"""DiT feature tower on MI355X: VAE encoder -> noisy latents -> DiT blocks -> 2x2-unfolded block output.

Device-side counterpart of `DiTFeaturizer.forward` (diffLVLM/src/models/dift_dit.py:168-196, pipeline :126-144, transformer
:18-124).  The VAE / noising half and all plumbing are `sd_engine.SdEngine`'s; the DiT core maps onto kernels the ViT path
already has, because the reference conditions on the TIMESTEP ONLY (dift_dit.py:9-16 drops the class embedding) and t is one
scalar per run - so every adaLN-Zero modulation is a constant vector per (t, block):
    LayerNorm(x) * (1 + scale) + shift          ->  visrep_layernorm with gamma = 1 + scale, beta = shift (no affine in the
                                                     checkpoint: normalization.py:69)
    gate * (attn_out / ff_out) + x              ->  GEMM epilogue EPI_RESID with its LayerScale vector = gate
    GELU(approximate="tanh") feed-forward       ->  GEMM epilogue EPI_ACT (gelu_tanh)
    2x2 patch embedding of the latents          ->  one GEMM (K = 2*2*8 zero-padded to 64) + sincos positions as the residual
Heads are 72 wide (DiT-XL/2: 16 x 72): zero-padded to 128 in the packed projections (csrc/attention.hip attn_fwd<2>).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from . import _lib
from .engine import gemm, layernorm, linear_vt
from .sd_engine import SdEngine, _ru, attention
from .sd_weights import DiTSpec


def _sincos(dim, grid_h, grid_w, base_size):
    # diffusers embeddings.py:70-128 get_2d_sincos_pos_embed (interpolation_scale 1; "w goes first" in the meshgrid)
    gh = np.arange(grid_h, dtype=np.float32) / (grid_h / base_size)
    gw = np.arange(grid_w, dtype=np.float32) / (grid_w / base_size)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_w, grid_h])

    def one(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    return torch.from_numpy(np.concatenate([one(dim // 2, grid[0]), one(dim // 2, grid[1])], axis=1)).float()


class DiTEngine(SdEngine):
    def __init__(self, spec: DiTSpec, w_dit: Dict[str, torch.Tensor], w_vae: Dict[str, torch.Tensor], device=None, up_ft_index: int = -1,
                 graph: bool = True):
        super().__init__(spec, w_dit, w_vae, device, up_ft_index, graph)
        self._ctx = {}                                   # no prompt: DiT is conditioned on the timestep only
        self._pos = {}

    def _check_spec(self):
        c = self.spec.core
        if c.d % 64:
            raise ValueError("DiT width must be a multiple of 64")
        if not -c.layers <= self.up_ft_index < c.layers:
            raise ValueError("up_ft_index out of range")
        self.n_blocks = self.up_ft_index + 1 if self.up_ft_index >= 0 else c.layers + self.up_ft_index + 1

    def _pack_core(self):
        c, w = self.spec.core, self.wu
        D, H = c.d, c.heads
        self.dp = _ru(c.head_dim, 64)
        pw = w["pos_embed.proj.weight"]                                           # [D, 4, p, p]
        W = torch.zeros(D, c.patch, c.patch, 8)
        W[..., : c.in_channels] = pw.permute(0, 2, 3, 1)                           # K order = (py, px, c8): matches the patch view
        self.P["patch"] = self._lin(W.reshape(D, -1), w["pos_embed.proj.bias"])
        pad_w = lambda n: self._pad_heads_out(w[n], H, self.dp)
        pad_b = lambda n: self._pad_heads_out(w[n][:, None], H, self.dp)[:, 0]
        for i in range(self.n_blocks):
            p = f"transformer_blocks.{i}"
            a = f"{p}.attn1"
            self.P[f"{p}.qk"] = self._lin(torch.cat([pad_w(f"{a}.to_q.weight"), pad_w(f"{a}.to_k.weight")], 0),
                                          torch.cat([pad_b(f"{a}.to_q.bias"), pad_b(f"{a}.to_k.bias")], 0))
            self.P[f"{p}.v"] = self._lin(pad_w(f"{a}.to_v.weight"), pad_b(f"{a}.to_v.bias"))
            self.P[f"{p}.o"] = self._lin(self._pad_heads_in(w[f"{a}.to_out.0.weight"], H, self.dp), w[f"{a}.to_out.0.bias"])
            self.P[f"{p}.ff1"] = self._lin(w[f"{p}.ff.net.0.proj.weight"], w[f"{p}.ff.net.0.proj.bias"])
            self.P[f"{p}.ff2"] = self._lin(w[f"{p}.ff.net.2.weight"], w[f"{p}.ff.net.2.bias"])

    def set_prompt(self, prompt_embeds):
        return None

    def set_timestep(self, t: int):
        """adaLN-Zero constants of every block for this t (normalization.py:81-84 with the timestep-only embedding)."""
        if self._t == int(t):
            return
        c, w = self.spec.core, self.wu
        half = 128
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - 1))      # downscale_freq_shift = 1
        e = float(t) * freqs
        tp = torch.cat([torch.cos(e), torch.sin(e)])[None]
        silu = torch.nn.functional.silu
        for i in range(self.n_blocks):
            p = f"transformer_blocks.{i}"
            em = f"{p}.norm1.emb.timestep_embedder"
            cond = silu(tp @ w[f"{em}.linear_1.weight"].t() + w[f"{em}.linear_1.bias"]) @ w[f"{em}.linear_2.weight"].t() + w[f"{em}.linear_2.bias"]
            mod = (silu(cond) @ w[f"{p}.norm1.linear.weight"].t() + w[f"{p}.norm1.linear.bias"])[0]
            shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6)
            f32 = lambda v: self._dev(v, torch.float32)
            self.P[f"{p}.mod"] = (f32(1 + scale_msa), f32(shift_msa), f32(gate_msa), f32(1 + scale_mlp), f32(shift_mlp), f32(gate_mlp))
        self._t = int(t)
        self._graphs.clear()

    def core_features(self, lat, B, H, W):
        """lat [B*H*W, 8] bf16 noisy latents -> [B, (gh/2)*(gw/2), 4*D] (dift_dit.py:191-195 + diffusion_encoder.py:84-88)."""
        c = self.spec.core
        D, ps = c.d, c.patch
        gh, gw = H // ps, W // ps
        N = gh * gw
        patches = lat.view(B, gh, ps, gw, ps, 8).permute(0, 1, 3, 2, 4, 5).reshape(B * N, ps * ps * 8)
        lin = self.P["patch"]
        cols = torch.zeros(B * N, lin.w.shape[1], dtype=torch.bfloat16, device=lat.device)
        cols[:, : patches.shape[1]] = patches
        key = (gh, gw, B)
        if key not in self._pos:
            # older entries stay: HIP graphs captured for another batch size hold their pointers (clearing here made a replay
            # after a batch-size change read freed memory - wrong features, then a memory fault under load)
            pos = _sincos(D, gh, gw, c.sample_size // ps).to(device=lat.device, dtype=torch.bfloat16)
            self._pos[key] = pos.repeat(B, 1).contiguous()
        h = gemm(cols, lin.w, lin.b, _lib.EPI_RESID, resid=self._pos[key])
        hd = c.heads * self.dp
        scale = c.head_dim ** -0.5
        for i in range(self.n_blocks):
            p = f"transformer_blocks.{i}"
            g1, b1, gate1, g3, b3, gate3 = self.P[f"{p}.mod"]
            n1 = layernorm(h, g1, b1, 1e-6)
            qk = gemm(n1, self.P[f"{p}.qk"].w, self.P[f"{p}.qk"].b)
            vt = linear_vt(n1, self.P[f"{p}.v"].w, self.P[f"{p}.v"].b)
            a = attention(qk[:, :hd], qk[:, hd:], vt, hd, B, N, N, c.heads, self.dp, scale, False)
            o = self.P[f"{p}.o"]
            gemm(a, o.w, o.b, _lib.EPI_RESID, resid=h, ls=gate1, out=h)
            n3 = layernorm(h, g3, b3, c.eps)
            f1, f2 = self.P[f"{p}.ff1"], self.P[f"{p}.ff2"]
            f = gemm(n3, f1.w, f1.b, _lib.EPI_ACT, act="gelu_tanh")
            gemm(f, f2.w, f2.b, _lib.EPI_RESID, resid=h, ls=gate3, out=h)
        # the reference's unfold sequence, verbatim layout ops on [B, N, D] (h = w = int(sqrt(N)), dift_dit.py:191)
        ft = h.view(B, N, D)
        s = int(N ** 0.5)
        ft = ft.transpose(2, 1).reshape(B, -1, s, s)
        ft = ft.unfold(3, 2, 2).unfold(2, 2, 2)
        ft = ft.reshape(B, -1, s // 2, s // 2, 4).permute(0, 4, 1, 2, 3).reshape(B, -1, s // 2, s // 2)
        return ft.permute(0, 2, 3, 1).reshape(B, (s // 2) * (s // 2), -1)

    @torch.no_grad()
    def forward(self, img, prompt_embeds=None, t: int = 1, ensemble_size: int = 1, post_noise=None, ddim_noise=None):
        if ensemble_size != 1:
            raise ValueError("dit does not enable ensemble (dift_dit.py:196)")
        return super().forward(img, None, t=t, ensemble_size=1, post_noise=post_noise, ddim_noise=ddim_noise)
