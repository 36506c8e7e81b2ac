"""SD3 (MMDiT) feature tower on MI355X: 16-channel VAE encoder -> flow-matching noise -> joint transformer blocks -> 2x2-
unfolded sample-stream output.

Device-side counterpart of `SD3Featurizer.forward` (diffLVLM/src/models/dift_sd3.py:139-175, pipeline :92-120, transformer
:10-91).  Same plan as dit_engine.DiTEngine: the conditioning vector is timestep embedding + pooled-prompt projection, i.e. a
constant per (t, prompt), so every adaLN-Zero / adaLN-continuous modulation of BOTH streams is folded into LayerNorm
gamma / beta vectors and GEMM LayerScale gates at `set_prompt` / `set_timestep`.  What is new is the JOINT attention
(attention_processor.py JointAttnProcessor2_0): image tokens and prompt tokens are projected by different weights and attend
as ONE sequence [image N | prompt L].  Per image, the two Q|K GEMMs write row ranges of one [N + L, 2*H*64] buffer and the two
V^T GEMMs write column ranges [0, N) / [N, N + L) of one V^T buffer (engine.linear_vt col_offset), so the flash kernel sees a
plain (N + L)-token self-attention and nothing is concatenated or copied.  Images are processed one at a time through the
blocks (their joint sequences are not 64-aligned to each other); the VAE half is batched.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import _lib
from .dit_engine import _sincos
from .engine import gemm, layernorm, linear_vt
from .sd_engine import SdEngine, _ru, attention
from .sd_weights import Sd3Spec


class Sd3Engine(SdEngine):
    def __init__(self, spec: Sd3Spec, w_core: Dict[str, torch.Tensor], w_vae: Dict[str, torch.Tensor], device=None, up_ft_index: int = -1,
                 graph: bool = True):
        super().__init__(spec, w_core, w_vae, device, up_ft_index, graph)
        self._pos = {}
        self._pooled = None

    def _check_spec(self):
        c = self.spec.core
        if c.d % 64 or c.joint_dim % 64 or c.pooled_dim % 8:
            raise ValueError("SD3 widths must be multiples of 64")
        if not -c.layers <= self.up_ft_index < c.layers:
            raise ValueError("up_ft_index out of range")
        self.n_blocks = self.up_ft_index + 1 if self.up_ft_index >= 0 else c.layers + self.up_ft_index + 1

    def noise_coefficients(self, t):
        """FlowMatchEulerDiscreteScheduler.add_noise as vendored and as called (dift_sd3.py:109-111): the RAW integer timestep."""
        return float(t), 1.0 - float(t)

    # ---------------------------------------------------------------- packing
    def _pack_core(self):
        c, w = self.spec.core, self.wu
        D, H = c.d, c.heads
        self.dp = _ru(c.head_dim, 64)
        zp = _ru(c.in_channels, 8)
        pw = w["pos_embed.proj.weight"]
        W = torch.zeros(D, c.patch, c.patch, zp)
        W[..., : c.in_channels] = pw.permute(0, 2, 3, 1)
        self.P["patch"] = self._lin(W.reshape(D, -1), w["pos_embed.proj.bias"])
        self.P["ctx_in"] = self._lin(w["context_embedder.weight"], w["context_embedder.bias"])
        pad_w = lambda n: self._pad_heads_out(w[n], H, self.dp)
        pad_b = lambda n: self._pad_heads_out(w[n][:, None], H, self.dp)[:, 0]
        for i in range(self.n_blocks):
            p = f"transformer_blocks.{i}"
            a = f"{p}.attn"
            last = i == c.layers - 1
            for tag, (q, k, v) in {"x": ("to_q", "to_k", "to_v"), "c": ("add_q_proj", "add_k_proj", "add_v_proj")}.items():
                self.P[f"{p}.qk_{tag}"] = self._lin(torch.cat([pad_w(f"{a}.{q}.weight"), pad_w(f"{a}.{k}.weight")], 0),
                                                    torch.cat([pad_b(f"{a}.{q}.bias"), pad_b(f"{a}.{k}.bias")], 0))
                self.P[f"{p}.v_{tag}"] = self._lin(pad_w(f"{a}.{v}.weight"), pad_b(f"{a}.{v}.bias"))
            self.P[f"{p}.o_x"] = self._lin(self._pad_heads_in(w[f"{a}.to_out.0.weight"], H, self.dp), w[f"{a}.to_out.0.bias"])
            streams = ["ff"]
            if not last:
                self.P[f"{p}.o_c"] = self._lin(self._pad_heads_in(w[f"{a}.to_add_out.weight"], H, self.dp), w[f"{a}.to_add_out.bias"])
                streams.append("ff_context")
            for ff in streams:
                self.P[f"{p}.{ff}1"] = self._lin(w[f"{p}.{ff}.net.0.proj.weight"], w[f"{p}.{ff}.net.0.proj.bias"])
                self.P[f"{p}.{ff}2"] = self._lin(w[f"{p}.{ff}.net.2.weight"], w[f"{p}.{ff}.net.2.bias"])

    # ---------------------------------------------------------------- per-run constants
    def set_prompt(self, prompt_embeds: torch.Tensor, pooled: Optional[torch.Tensor] = None):
        """prompt_embeds [1, L, joint_dim], pooled [1, pooled_dim] (pipe.encode_prompt, dift_sd3.py:152-158)."""
        if pooled is None:
            raise ValueError("SD3 needs the pooled prompt projections next to the prompt embeddings")
        pe = prompt_embeds.reshape(-1, prompt_embeds.shape[-1]).to(device=self.device, dtype=torch.bfloat16).contiguous()
        lin = self.P["ctx_in"]
        self._ctx = {"c0": gemm(pe, lin.w, lin.b)}                               # context_embedder: [L, D]
        self._ctx_len = pe.shape[0]
        self._pooled = pooled.detach().float().cpu().reshape(1, -1)
        self._ctx_version += 1
        self._graphs.clear()
        if self._t is not None:
            t, self._t = self._t, None
            self.set_timestep(t)

    def set_timestep(self, t: int):
        if self._pooled is None:
            self._t = int(t)                                                      # folded once the prompt is known
            return
        if self._t == int(t) and "mods_ready" in self.P:
            return
        c, w = self.spec.core, self.wu
        half = 128
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)            # downscale_freq_shift = 0
        e = float(t) * freqs
        tp = torch.cat([torch.cos(e), torch.sin(e)])[None]
        silu = torch.nn.functional.silu
        lin = lambda v, n: v @ w[f"time_text_embed.{n}.weight"].t() + w[f"time_text_embed.{n}.bias"]
        temb = (lin(silu(lin(tp, "timestep_embedder.linear_1")), "timestep_embedder.linear_2")
                + lin(silu(lin(self._pooled, "text_embedder.linear_1")), "text_embedder.linear_2"))
        act = silu(temb)
        f32 = lambda v: self._dev(v, torch.float32)
        for i in range(self.n_blocks):
            p = f"transformer_blocks.{i}"
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = (act @ w[f"{p}.norm1.linear.weight"].t() + w[f"{p}.norm1.linear.bias"])[0].chunk(6)
            self.P[f"{p}.mod_x"] = (f32(1 + sc_a), f32(sh_a), f32(g_a), f32(1 + sc_m), f32(sh_m), f32(g_m))
            cm = (act @ w[f"{p}.norm1_context.linear.weight"].t() + w[f"{p}.norm1_context.linear.bias"])[0]
            if i == c.layers - 1:                                                 # AdaLayerNormContinuous: (scale, shift)
                sc, sh = cm.chunk(2)
                self.P[f"{p}.mod_c"] = (f32(1 + sc), f32(sh))
            else:
                c_sh_a, c_sc_a, c_g_a, c_sh_m, c_sc_m, c_g_m = cm.chunk(6)
                self.P[f"{p}.mod_c"] = (f32(1 + c_sc_a), f32(c_sh_a), f32(c_g_a), f32(1 + c_sc_m), f32(c_sh_m), f32(c_g_m))
        self.P["mods_ready"] = True
        self._t = int(t)
        self._graphs.clear()

    # ---------------------------------------------------------------- MMDiT core
    def _ff(self, x, p, ff, gamma, beta, gate):
        n = layernorm(x, gamma, beta, 1e-6)
        f1, f2 = self.P[f"{p}.{ff}1"], self.P[f"{p}.{ff}2"]
        f = gemm(n, f1.w, f1.b, _lib.EPI_ACT, act="gelu_tanh")
        gemm(f, f2.w, f2.b, _lib.EPI_RESID, resid=x, ls=gate, out=x)

    def core_features(self, lat, B, H, W):
        """lat [B*H*W, 16] bf16 noisy latents -> [B, (gh/2)*(gw/2), 4*D] (dift_sd3.py:170-174 + diffusion_encoder.py:84-88)."""
        if "mods_ready" not in self.P:
            raise RuntimeError("set_timestep() and set_prompt() must be called before the transformer runs")
        c = self.spec.core
        D, ps, heads, dp = c.d, c.patch, c.heads, self.dp
        gh, gw = H // ps, W // ps
        N, L = gh * gw, self._ctx_len
        if N % 16:
            raise NotImplementedError("the joint V^T layout needs the image-token count to be a multiple of 16")
        zp = lat.shape[1]
        patches = lat.view(B, gh, ps, gw, ps, zp).permute(0, 1, 3, 2, 4, 5).reshape(B * N, ps * ps * zp)
        lin = self.P["patch"]
        if patches.shape[1] != lin.w.shape[1]:
            cols = torch.zeros(B * N, lin.w.shape[1], dtype=torch.bfloat16, device=lat.device)
            cols[:, : patches.shape[1]] = patches
            patches = cols
        key = (gh, gw, B)
        if key not in self._pos:
            # older entries stay: HIP graphs captured for another batch size hold their pointers (clearing here made a replay
            # after a batch-size change read freed memory - wrong features, then a memory fault under load)
            full = _sincos(D, c.pos_max, c.pos_max, c.sample_size // ps).reshape(c.pos_max, c.pos_max, D)
            top, left = (c.pos_max - gh) // 2, (c.pos_max - gw) // 2
            pos = full[top: top + gh, left: left + gw].reshape(N, D).to(device=lat.device, dtype=torch.bfloat16)
            self._pos[key] = pos.repeat(B, 1).contiguous()
        h_all = gemm(patches.contiguous(), lin.w, lin.b, _lib.EPI_RESID, resid=self._pos[key])
        hd, T = heads * dp, N + L
        scale = c.head_dim ** -0.5
        ldvt = _ru(T, 64) + 64
        for b in range(B):
            x = h_all[b * N:(b + 1) * N]
            ctx = self._ctx["c0"].clone()
            for i in range(self.n_blocks):
                p = f"transformer_blocks.{i}"
                last = i == c.layers - 1
                mx, mc = self.P[f"{p}.mod_x"], self.P[f"{p}.mod_c"]
                nx = layernorm(x, mx[0], mx[1], 1e-6)
                nc = layernorm(ctx, mc[0], mc[1], 1e-6)
                qk = torch.empty(T, 2 * hd, dtype=torch.bfloat16, device=lat.device)
                vt = torch.zeros(hd, ldvt, dtype=torch.bfloat16, device=lat.device)
                for tag, src, r0 in (("x", nx, 0), ("c", nc, N)):
                    lq, lv = self.P[f"{p}.qk_{tag}"], self.P[f"{p}.v_{tag}"]
                    gemm(src, lq.w, lq.b, out=qk[r0: r0 + src.shape[0]])
                    linear_vt(src, lv.w, lv.b, out=vt, col_offset=r0)
                a = attention(qk[:, :hd], qk[:, hd:], vt, hd, 1, T, T, heads, dp, scale, False)
                o = self.P[f"{p}.o_x"]
                gemm(a[:N], o.w, o.b, _lib.EPI_RESID, resid=x, ls=mx[2], out=x)
                self._ff(x, p, "ff", mx[3], mx[4], mx[5])
                if not last:
                    o = self.P[f"{p}.o_c"]
                    gemm(a[N:], o.w, o.b, _lib.EPI_RESID, resid=ctx, ls=mc[2], out=ctx)
                    self._ff(ctx, p, "ff_context", mc[3], mc[4], mc[5])
        ft = h_all.view(B, N, D)
        s = int(N ** 0.5)                                                         # the reference's unfold, verbatim layout ops
        ft = ft.transpose(2, 1).reshape(B, -1, s, s)
        ft = ft.unfold(3, 2, 2).unfold(2, 2, 2)
        ft = ft.reshape(B, -1, s // 2, s // 2, 4).permute(0, 4, 1, 2, 3).reshape(B, -1, s // 2, s // 2)
        return ft.permute(0, 2, 3, 1).reshape(B, (s // 2) * (s // 2), -1)

    @torch.no_grad()
    def forward(self, img, prompt_embeds=None, t: int = 1, ensemble_size: int = 1, post_noise=None, ddim_noise=None, pooled=None):
        if ensemble_size != 1:
            raise ValueError("dit does not enable ensemble (dift_sd3.py:175)")
        if prompt_embeds is not None and prompt_embeds is not self._prompt_src:
            self.set_prompt(prompt_embeds, pooled)
            self._prompt_src = prompt_embeds
        return super().forward(img, None, t=t, ensemble_size=1, post_noise=post_noise, ddim_noise=ddim_noise)
