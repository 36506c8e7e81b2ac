"""ViT tower oracle (CPU, fp32).  TEST INFRASTRUCTURE — see oracle/__init__.py.

The reference towers are thin wrappers over HuggingFace `transformers` vision
models (third-party, NOT under /root/reference; pinned 4.31.0 / 4.38.2 by the
reference, 5.15.0 installed here — SURVEY.md F12):

  * CLIPVisionTower.forward      llava/model/multimodal_encoder/clip_encoder.py:39-51
      -> CLIPVisionModel(pixels, output_hidden_states=True).hidden_states[select_layer]
      -> feature_select 'patch' drops CLS              clip_encoder.py:29-37
  * DinoV2VisionTower.forward    llava/model/multimodal_encoder/dinov2_encoder.py:42-54
  * SigLipVisionTower.forward    llava/model/multimodal_encoder/siglip_encoder.py:40-52

This file restates the published pre-LN ViT arithmetic those models run:
patch conv (k = s = patch) -> [CLS] concat -> + position embedding ->
(CLIP: pre_layrnorm) -> L x { x += ls1 * Attn(LN1(x)); x += ls2 * MLP(LN2(x)) }
and returns every hidden state so `hidden_states[select_layer]` is available.
It is pinned against the HF modules themselves (random-init, tiny and full-size
configs) by tests/golden/make_golden.py; see tests/golden/vit_*.npz.

Weights are consumed in the package's packed layout
(law_of_vision_representation_in_mllms_amd.vit_weights.pack_*): a dict with
  patch_w [d, 3*p*p], patch_b [d]|None, cls [d]|None, pos [T, d],
  pre_ln_g/pre_ln_b [d]|None,
  layers: list of dicts ln1_g ln1_b wqkv[3d,d] bqkv[3d] wo[d,d] bo[d] ls1|None
                         ln2_g ln2_b w1[m,d] b1[m] w2[d,m] b2[d] ls2|None
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _act(x: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "quick_gelu":            # HF activations.QuickGELUActivation
        return x * torch.sigmoid(1.702 * x)
    if kind == "gelu":                  # exact erf
        return F.gelu(x)
    if kind in ("gelu_tanh", "gelu_pytorch_tanh"):
        return F.gelu(x, approximate="tanh")
    raise ValueError(kind)


def vit_hidden_states(spec, w, pixels: torch.Tensor, n_layers: int | None = None,
                      dtype=torch.float32):
    """Return [h0, h1, ..., h_n] like HF `output_hidden_states=True`.

    spec: object/dict with patch, d, heads, act, eps (see package ViTSpec).
    pixels: [B, 3, H, W].
    """
    g = (lambda k: getattr(spec, k)) if not isinstance(spec, dict) else spec.__getitem__
    p, d, heads, act, eps = g("patch"), g("d"), g("heads"), g("act"), g("eps")
    c = lambda t: None if t is None else t.to(dtype)
    x = pixels.to(dtype)
    B = x.shape[0]
    pw = c(w["patch_w"]).view(d, 3, p, p)
    t = F.conv2d(x, pw, c(w.get("patch_b")), stride=p)           # [B, d, gh, gw]
    t = t.flatten(2).transpose(1, 2)                              # [B, P, d]
    if w.get("cls") is not None:
        t = torch.cat([c(w["cls"]).view(1, 1, d).expand(B, 1, d), t], dim=1)
    t = t + c(w["pos"]).unsqueeze(0)
    if w.get("pre_ln_g") is not None:
        t = F.layer_norm(t, (d,), c(w["pre_ln_g"]), c(w["pre_ln_b"]), eps)
    hs = [t]
    layers = w["layers"] if n_layers is None else w["layers"][:n_layers]
    dh = d // heads
    scale = 1.0 / math.sqrt(dh)
    for L in layers:
        h = F.layer_norm(t, (d,), c(L["ln1_g"]), c(L["ln1_b"]), eps)
        qkv = F.linear(h, c(L["wqkv"]), c(L["bqkv"]))             # [B, T, 3d]
        q, k, v = qkv.split(d, dim=-1)
        T = q.shape[1]
        q = q.view(B, T, heads, dh).transpose(1, 2)
        k = k.view(B, T, heads, dh).transpose(1, 2)
        v = v.view(B, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * scale
        a = torch.softmax(s.float(), dim=-1).to(dtype)
        o = (a @ v).transpose(1, 2).reshape(B, T, d)
        o = F.linear(o, c(L["wo"]), c(L["bo"]))
        if L.get("ls1") is not None:
            o = o * c(L["ls1"])
        t = t + o
        h = F.layer_norm(t, (d,), c(L["ln2_g"]), c(L["ln2_b"]), eps)
        h = _act(F.linear(h, c(L["w1"]), c(L["b1"])), act)
        h = F.linear(h, c(L["w2"]), c(L["b2"]))
        if L.get("ls2") is not None:
            h = h * c(L["ls2"])
        t = t + h
        hs.append(t)
    return hs


def tower_features(spec, w, pixels, select_layer=-2, select_feature="patch", dtype=torch.float32):
    """feature_select semantics of clip_encoder.py:29-37 over the oracle hidden states."""
    total = len(w["layers"])
    idx = select_layer if select_layer >= 0 else total + 1 + select_layer
    hs = vit_hidden_states(spec, w, pixels, n_layers=idx, dtype=dtype)
    f = hs[idx]
    has_cls = w.get("cls") is not None
    if select_feature == "patch":
        f = f[:, 1:]                     # the reference slices index 0 whether or not it is a CLS
    elif select_feature != "cls_patch":
        raise ValueError(f"Unexpected select feature: {select_feature}")
    return f


def clip_image_embeds(spec, w, post_ln_g, post_ln_b, projection, pixels, dtype=torch.float32):
    """HF CLIPVisionModelWithProjection(pixels).image_embeds (modeling_clip.py CLIPVisionTransformer.forward pooled_output
    = post_layernorm(last_hidden_state[:, 0]); then visual_projection, no bias) - the image-variation pipeline's
    `_encode_image` (vendored pipeline_stable_diffusion_image_variation.py:133-141)."""
    hs = vit_hidden_states(spec, w, pixels, dtype=dtype)
    cls = hs[-1][:, 0].to(dtype)
    n = torch.nn.functional.layer_norm(cls, (spec.d,), post_ln_g.to(dtype), post_ln_b.to(dtype), spec.eps)
    return n @ projection.to(dtype).t()
