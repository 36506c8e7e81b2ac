"""CLIP text encoder on MI355X: the prompt embeddings of the diffusion towers (`pipe.encode_prompt`, dift_sd.py:258-263).

HF CLIPTextTransformer (modeling_clip.py): token + position embeddings -> causal-masked pre-LN encoder -> final LN.
One 77-token sequence per prompt, computed once per tower and reused for every image, so this is composed in Python over
the C-ABI primitives (GEMM with fused bias / activation / residual, LayerNorm, causal attention); head width must be 64
(CLIP ViT-L/14 text: 768 / 12, OpenCLIP ViT-H text of SD2.1: 1024 / 16).
"""
from __future__ import annotations

import torch

from . import _lib
from .engine import gemm, layernorm, linear_vt
from .sd_engine import attention
from .sd_weights import TextSpec


class ClipTextEngine:
    def __init__(self, spec: TextSpec, weights, device=None):
        _lib.require_gpu()
        if spec.d != spec.heads * 64:
            raise ValueError("CLIP text encoder: head width must be 64")
        self.spec = spec
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        mat = lambda t: t.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()
        vec = lambda t: t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        w = weights
        self.tok = vec(w["embeddings.token_embedding.weight"])
        self.pos = vec(w["embeddings.position_embedding.weight"])
        self.layers = []
        for i in range(spec.layers):
            p = f"encoder.layers.{i}"
            a = f"{p}.self_attn"
            self.layers.append(dict(
                ln1=(vec(w[f"{p}.layer_norm1.weight"]), vec(w[f"{p}.layer_norm1.bias"])),
                ln2=(vec(w[f"{p}.layer_norm2.weight"]), vec(w[f"{p}.layer_norm2.bias"])),
                wqk=mat(torch.cat([w[f"{a}.q_proj.weight"], w[f"{a}.k_proj.weight"]], 0)),
                bqk=vec(torch.cat([w[f"{a}.q_proj.bias"], w[f"{a}.k_proj.bias"]], 0)),
                wv=mat(w[f"{a}.v_proj.weight"]), bv=vec(w[f"{a}.v_proj.bias"]),
                wo=mat(w[f"{a}.out_proj.weight"]), bo=vec(w[f"{a}.out_proj.bias"]),
                w1=mat(w[f"{p}.mlp.fc1.weight"]), b1=vec(w[f"{p}.mlp.fc1.bias"]),
                w2=mat(w[f"{p}.mlp.fc2.weight"]), b2=vec(w[f"{p}.mlp.fc2.bias"])))
        self.final = (vec(w["final_layer_norm.weight"]), vec(w["final_layer_norm.bias"]))

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, hidden_state=None) -> torch.Tensor:
        """input_ids [B, L] int64 -> last_hidden_state [B, L, d] bf16; hidden_state=-2 -> HF `hidden_states[-2]` (second-to-
        last layer's output, no final LayerNorm), which SDXL's encode_prompt concatenates from its two encoders."""
        s = self.spec
        B, L = input_ids.shape
        if L > s.max_pos:
            raise ValueError(f"sequence length {L} exceeds max_position_embeddings {s.max_pos}")
        ids = input_ids.to(self.device)
        h = (self.tok[ids] + self.pos[:L][None]).to(torch.bfloat16).reshape(B * L, s.d).contiguous()   # embedding gather
        d = s.d
        layers = self.layers if hidden_state is None else self.layers[: len(self.layers) + 1 + hidden_state]
        for P in layers:
            n1 = layernorm(h, *P["ln1"], s.eps)
            qk = gemm(n1, P["wqk"], P["bqk"])
            vt = linear_vt(n1, P["wv"], P["bv"])
            a = attention(qk[:, :d], qk[:, d:], vt, d, B, L, L, s.heads, 64, 0.125, False, causal=True)
            gemm(a, P["wo"], P["bo"], _lib.EPI_RESID, resid=h, out=h)
            n2 = layernorm(h, *P["ln2"], s.eps)
            f = gemm(n2, P["w1"], P["b1"], _lib.EPI_ACT, act=s.act)
            gemm(f, P["w2"], P["b2"], _lib.EPI_RESID, resid=h, out=h)
        if hidden_state is not None:
            return h.view(B, L, d)
        return layernorm(h, *self.final, s.eps).view(B, L, d)
