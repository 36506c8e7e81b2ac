"""ViT tower specs and weight packing (host side, torch-CPU only — no device work here).

The reference towers delegate the arithmetic to HuggingFace vision models
(clip_encoder.py:24, dinov2_encoder.py:27, siglip_encoder.py:25).  The MI355X
engine consumes ONE packed layout for all of them; this module converts HF
`state_dict`s into it and can also synthesise weights with a version-stable
PRNG (`numpy.random.RandomState`) when no checkpoint is available offline.

Packed layout (fp32 torch tensors on CPU):
  patch_w [d, 3*p*p]  patch_b [d]|None  cls [d]|None  pos [T, d]
  pre_ln_g / pre_ln_b [d]|None
  layers[i]: ln1_g ln1_b wqkv[3d,d] bqkv[3d] wo[d,d] bo[d] ls1[d]|None
             ln2_g ln2_b w1[m,d] b1[m] w2[d,m] b2[d] ls2[d]|None
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Dict, Optional

import numpy as np
import torch


@dataclass(frozen=True)
class ViTSpec:
    name: str
    image_size: int
    patch: int
    d: int
    layers: int
    heads: int
    mlp: int
    act: str            # quick_gelu | gelu | gelu_tanh
    eps: float
    has_cls: bool
    pre_ln: bool        # CLIP pre_layrnorm
    patch_bias: bool
    layerscale: bool    # DINOv2
    family: str         # clip | dinov2 | siglip
    pos_grid: int = 0   # native position-embedding grid (DINOv2: 37); 0 = image_size // patch

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def tokens(self) -> int:
        return self.num_patches + (1 if self.has_cls else 0)

    def at_resolution(self, image_size: int) -> "ViTSpec":
        return replace(self, image_size=image_size)


# Backbones behind the reference registry (llava_arch.py:29-40; shapes from the public HF configs)
SPECS: Dict[str, ViTSpec] = {
    "openai/clip-vit-large-patch14-336": ViTSpec("openai/clip-vit-large-patch14-336", 336, 14, 1024, 24, 16, 4096,
                                                 "quick_gelu", 1e-5, True, True, False, False, "clip"),
    "openai/clip-vit-large-patch14": ViTSpec("openai/clip-vit-large-patch14", 224, 14, 1024, 24, 16, 4096,
                                             "quick_gelu", 1e-5, True, True, False, False, "clip"),
    "laion/CLIP-ViT-L-14-laion2B-s32B-b82K": ViTSpec("laion/CLIP-ViT-L-14-laion2B-s32B-b82K", 224, 14, 1024, 24, 16,
                                                     4096, "gelu", 1e-5, True, True, False, False, "clip"),
    "facebook/dinov2-large": ViTSpec("facebook/dinov2-large", 224, 14, 1024, 24, 16, 4096,
                                     "gelu", 1e-6, True, False, True, True, "dinov2", pos_grid=37),
    "google/siglip-base-patch16-224": ViTSpec("google/siglip-base-patch16-224", 224, 16, 768, 12, 12, 3072,
                                              "gelu_tanh", 1e-6, False, False, True, False, "siglip"),
}


def tiny_spec(family: str, act: Optional[str] = None, image_size: int = 28, patch: int = 7, d: int = 64,
              layers: int = 3, heads: int = 2, mlp: int = 128) -> ViTSpec:
    """Small spec of a given family for fixtures / tests."""
    base = {"clip": SPECS["openai/clip-vit-large-patch14"], "dinov2": SPECS["facebook/dinov2-large"],
            "siglip": SPECS["google/siglip-base-patch16-224"]}[family]
    return replace(base, name=f"tiny-{family}", image_size=image_size, patch=patch, d=d, layers=layers,
                   heads=heads, mlp=mlp, act=act or base.act, pos_grid=0)


def spec_from_hf_config(cfg, name: str = "", crop_size: Optional[int] = None) -> ViTSpec:
    """Build a spec from a HF CLIPVisionConfig / Dinov2Config / SiglipVisionConfig.

    DINOv2: `cfg.image_size` (518 for facebook/dinov2-*) is the NATIVE position-embedding grid (37 x 37), not the input size the
    reference runs: its AutoImageProcessor resizes to 256 and centre-crops to `crop_size` (224 by default,
    dinov2_encoder.py:24,44-47) and HF interpolates the position embedding down to that grid.  So the tower is built at the
    processor's crop size (`crop_size`, read from preprocessor_config.json by the caller; 224 when absent) and only `pos_grid`
    keeps the native size."""
    mt = getattr(cfg, "model_type", "")
    if hasattr(cfg, "vision_config") and mt in ("clip", "siglip"):
        cfg = cfg.vision_config
        mt = cfg.model_type
    act = {"quick_gelu": "quick_gelu", "gelu": "gelu", "gelu_pytorch_tanh": "gelu_tanh", "gelu_new": "gelu_tanh",
           "gelu_tanh": "gelu_tanh"}[getattr(cfg, "hidden_act", "gelu")]
    if mt.startswith("clip"):
        return ViTSpec(name, cfg.image_size, cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                       cfg.num_attention_heads, cfg.intermediate_size, act, cfg.layer_norm_eps,
                       True, True, False, False, "clip")
    if mt.startswith("dinov2"):
        return ViTSpec(name, int(crop_size or 224), cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                       cfg.num_attention_heads, int(cfg.hidden_size * cfg.mlp_ratio), act, cfg.layer_norm_eps,
                       True, False, True, True, "dinov2", pos_grid=cfg.image_size // cfg.patch_size)
    if mt.startswith("siglip"):
        return ViTSpec(name, cfg.image_size, cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                       cfg.num_attention_heads, cfg.intermediate_size, act, cfg.layer_norm_eps,
                       False, False, True, False, "siglip")
    raise ValueError(f"unsupported HF model_type {mt!r}")


def _strip(sd: Dict[str, torch.Tensor], prefixes) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v.detach().float().cpu()
    return out


def interpolate_pos(pos: torch.Tensor, has_cls: bool, grid: int) -> torch.Tensor:
    """Bicubic position-embedding resize (HF Dinov2Embeddings.interpolate_pos_encoding, size= form, fp32)."""
    cls = pos[:1] if has_cls else pos[:0]
    patch = pos[1:] if has_cls else pos
    n = int(round(patch.shape[0] ** 0.5))
    if n == grid:
        return pos
    d = pos.shape[-1]
    pp = patch.reshape(1, n, n, d).permute(0, 3, 1, 2).float()
    pp = torch.nn.functional.interpolate(pp, size=(grid, grid), mode="bicubic", align_corners=False)
    pp = pp.permute(0, 2, 3, 1).reshape(grid * grid, d)
    return torch.cat([cls, pp], dim=0)


def pack_hf_state_dict(sd: Dict[str, torch.Tensor], spec: ViTSpec) -> dict:
    """HF CLIPVisionModel / Dinov2Model / SiglipVisionModel(.vision_model) state_dict -> packed layout."""
    d = spec.d
    if spec.family == "clip":
        s = _strip(sd, ["vision_model."])
        w = {
            "patch_w": s["embeddings.patch_embedding.weight"].reshape(d, -1),
            "patch_b": None,
            "cls": s["embeddings.class_embedding"].reshape(d),
            "pos": s["embeddings.position_embedding.weight"],
            "pre_ln_g": s["pre_layrnorm.weight"], "pre_ln_b": s["pre_layrnorm.bias"],
        }
        lay = lambda i, k: s[f"encoder.layers.{i}.{k}"]
        names = dict(q="self_attn.q_proj", k="self_attn.k_proj", v="self_attn.v_proj", o="self_attn.out_proj",
                     ln1="layer_norm1", ln2="layer_norm2", fc1="mlp.fc1", fc2="mlp.fc2")
    elif spec.family == "siglip":
        s = _strip(sd, ["vision_model."])
        w = {
            "patch_w": s["embeddings.patch_embedding.weight"].reshape(d, -1),
            "patch_b": s["embeddings.patch_embedding.bias"],
            "cls": None,
            "pos": s["embeddings.position_embedding.weight"],
            "pre_ln_g": None, "pre_ln_b": None,
        }
        lay = lambda i, k: s[f"encoder.layers.{i}.{k}"]
        names = dict(q="self_attn.q_proj", k="self_attn.k_proj", v="self_attn.v_proj", o="self_attn.out_proj",
                     ln1="layer_norm1", ln2="layer_norm2", fc1="mlp.fc1", fc2="mlp.fc2")
    elif spec.family == "dinov2":
        s = _strip(sd, ["dinov2."])
        w = {
            "patch_w": s["embeddings.patch_embeddings.projection.weight"].reshape(d, -1),
            "patch_b": s["embeddings.patch_embeddings.projection.bias"],
            "cls": s["embeddings.cls_token"].reshape(d),
            "pos": s["embeddings.position_embeddings"].reshape(-1, d),
            "pre_ln_g": None, "pre_ln_b": None,
        }
        lay = lambda i, k: s[f"encoder.layer.{i}.{k}"]
        names = dict(q="attention.attention.query", k="attention.attention.key", v="attention.attention.value",
                     o="attention.output.dense", ln1="norm1", ln2="norm2", fc1="mlp.fc1", fc2="mlp.fc2")
    else:
        raise ValueError(spec.family)
    # The checkpoint's own grid stays in the packed weights: a later change of resolution (weights_at_resolution) must resize from
    # IT, once, like HF's interpolate_pos_encoding does (37 x 37 -> target), not from the already-resized table (37 -> 16 -> 24).
    w["pos_native"] = w["pos"]
    w["pos"] = interpolate_pos(w["pos"], spec.has_cls, spec.grid)
    layers = []
    for i in range(spec.layers):
        L = {
            "ln1_g": lay(i, names["ln1"] + ".weight"), "ln1_b": lay(i, names["ln1"] + ".bias"),
            "wqkv": torch.cat([lay(i, names[c] + ".weight") for c in "qkv"], dim=0),
            "bqkv": torch.cat([lay(i, names[c] + ".bias") for c in "qkv"], dim=0),
            "wo": lay(i, names["o"] + ".weight"), "bo": lay(i, names["o"] + ".bias"),
            "ln2_g": lay(i, names["ln2"] + ".weight"), "ln2_b": lay(i, names["ln2"] + ".bias"),
            "w1": lay(i, names["fc1"] + ".weight"), "b1": lay(i, names["fc1"] + ".bias"),
            "w2": lay(i, names["fc2"] + ".weight"), "b2": lay(i, names["fc2"] + ".bias"),
            "ls1": None, "ls2": None,
        }
        if spec.layerscale:
            L["ls1"] = lay(i, "layer_scale1.lambda1")
            L["ls2"] = lay(i, "layer_scale2.lambda1")
        layers.append(L)
    w["layers"] = layers
    return w


def synthetic_weights(spec: ViTSpec, seed: int = 1, n_layers: Optional[int] = None) -> dict:
    """Deterministic random-init weights (numpy RandomState: stream frozen across versions).

    Scales are chosen so activations stay O(1) through 24 layers (std 0.02 matrices,
    LN gains around 1, LayerScale around 0.1 for DINOv2-style towers).
    """
    rs = np.random.RandomState(seed)
    d, m, p = spec.d, spec.mlp, spec.patch
    import os
    mode = os.environ.get("VISREP_FAST_SYNTHETIC")                     # throughput runs only: torch generator ("1") / drawn in HBM ("cuda")
    fdev = "cuda" if mode == "cuda" and torch.cuda.is_available() else "cpu"
    fast = torch.Generator(device=fdev).manual_seed(seed) if mode in ("1", "cuda") else None

    def rn(*shape, std=0.02, mean=0.0):
        if fast is not None:
            return (torch.randn(shape, generator=fast, device=fdev) * std + mean).cpu()
        return torch.from_numpy((rs.standard_normal(shape) * std + mean).astype(np.float32))

    w = {
        "patch_w": rn(d, 3 * p * p), "patch_b": rn(d) if spec.patch_bias else None,
        "cls": rn(d) if spec.has_cls else None, "pos": rn(spec.tokens, d),
        "pre_ln_g": rn(d, std=0.1, mean=1.0) if spec.pre_ln else None,
        "pre_ln_b": rn(d) if spec.pre_ln else None,
    }
    layers = []
    for _ in range(spec.layers if n_layers is None else n_layers):
        layers.append({
            "ln1_g": rn(d, std=0.1, mean=1.0), "ln1_b": rn(d),
            "wqkv": rn(3 * d, d), "bqkv": rn(3 * d),
            "wo": rn(d, d), "bo": rn(d),
            "ls1": rn(d, std=0.02, mean=0.1) if spec.layerscale else None,
            "ln2_g": rn(d, std=0.1, mean=1.0), "ln2_b": rn(d),
            "w1": rn(m, d), "b1": rn(m), "w2": rn(d, m), "b2": rn(d),
            "ls2": rn(d, std=0.02, mean=0.1) if spec.layerscale else None,
        })
    w["layers"] = layers
    return w


def flatten(w: dict) -> Dict[str, torch.Tensor]:
    """Packed dict -> flat {name: tensor} (for npz fixtures)."""
    out = {k: v for k, v in w.items() if k != "layers" and v is not None}
    for i, L in enumerate(w["layers"]):
        for k, v in L.items():
            if v is not None:
                out[f"layers.{i}.{k}"] = v
    return out


def unflatten(flat: Dict[str, torch.Tensor]) -> dict:
    w: dict = {k: None for k in ("patch_b", "cls", "pre_ln_g", "pre_ln_b")}
    layers: Dict[int, dict] = {}
    for k, v in flat.items():
        v = torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v
        if k.startswith("layers."):
            _, i, name = k.split(".", 2)
            layers.setdefault(int(i), {"ls1": None, "ls2": None})[name] = v
        else:
            w[k] = v
    w["layers"] = [layers[i] for i in sorted(layers)]
    return w


def weights_at_resolution(spec: ViTSpec, w: dict, image_size: int):
    """Return (spec', w') for another input resolution: only the position embedding changes (bicubic resize)."""
    spec2 = spec.at_resolution(image_size)
    w2 = dict(w)
    src = w.get("pos_native")
    if src is None:
        src = w["pos"]
    w2["pos_native"] = src                                   # every later resize starts from the same table
    w2["pos"] = interpolate_pos(src, spec.has_cls, spec2.grid)
    return spec2, w2


def hostile_weights(spec: ViTSpec, seed: int = 1, n_layers: Optional[int] = None, profile: str = "outlier", n_outlier: int = 4,
                    outlier=(50.0, 200.0), ls_range=(1e-5, 1.0), const_token: int = 7, const_value: float = 25.0, sharp: float = 5.0) -> dict:
    """Synthetic weights in the regime real checkpoints live in and N(0, 0.02) draws do not: the stand-in for the
    CLIP / DINOv2 checkpoints that are not available offline.  Starts from `synthetic_weights` and plants, deterministically per seed:

    profile "outlier" (and "all"):
      * massive-activation channels: `n_outlier` residual channels whose writers - the rows of one early layer's fc2 (`w2`) and of one middle
        layer's attention output (`wo`) - are scaled by log-uniform factors in `outlier` (50-200x) and whose writer biases are set to
        +-(50-200) in units of the stream's O(1) scale: those channels carry values 50-200x the others from that layer on, for every token
        (a token-dependent and a token-independent part, as the massive activations of ViT-L checkpoints have);
      * LayerNorm gains that fight them: in every later layer half of those channels get gamma = 0.02 and half gamma = 3 (ln1 and ln2 the
        other way round), so the folded GEMM (gamma o W) sees both a squashed and an amplified outlier column.
    profile "const" (and "all"):
      * one near-constant token: `const_value` (25) added to every channel of position-embedding row `const_token` (a patch token).  Towers
        without a pre-LayerNorm (DINOv2, SigLIP) carry that row at |mean| ~ 25-50 std through every layer - the case E[x^2] - mean^2 row
        statistics are worst at; CLIP's pre_layrnorm removes the shift (there the pre-LN's own statistics pass is what it exercises).
        (With the outlier channels present the row's std is theirs, |mean| / std drops to ~2-5: hence a profile of its own.)
    every profile:
      * LayerScale (DINOv2 family) spread log-uniformly over `ls_range` (1e-5 .. 1) with random signs (outlier channels keep 0.5 so that
        the planted activations survive);
      * sharp attention heads: the Q and K rows of two heads scaled by `sharp` (5) in every third layer - logits x25: near-one-hot softmax
        rows with large pre-softmax scores (the pre-scaled-Q / running-reference paths of the attention kernels).
    Returns the packed dict plus w["hostile"] = {"profile", "channels", "factors", "const_token", ...} for the tests' own assertions."""
    if profile not in ("outlier", "const", "all"):
        raise ValueError(f"profile must be 'outlier', 'const' or 'all', got {profile!r}")
    w = synthetic_weights(spec, seed=seed, n_layers=n_layers)
    rs = np.random.RandomState(seed + 7919)
    d, n = spec.d, len(w["layers"])
    ch = np.sort(rs.choice(d, n_outlier, replace=False))
    fac = np.exp(rs.uniform(np.log(outlier[0]), np.log(outlier[1]), n_outlier)).astype(np.float32)
    sgn_o = np.where(rs.rand(n_outlier) < 0.5, -1.0, 1.0).astype(np.float32)
    early, mid = min(1, n - 1), min(max(n // 3, 1), n - 1)
    half = max(n_outlier // 2, 1)
    cht, fact = torch.from_numpy(ch), torch.from_numpy(fac)
    plant = profile in ("outlier", "all")
    lsdiv = 0.5 if spec.layerscale else 1.0
    if plant:
        for L, wk, bk, sl in ((w["layers"][early], "w2", "b2", slice(0, half)), (w["layers"][mid], "wo", "bo", slice(half, None))):
            L[wk] = L[wk].clone()
            L[wk][cht[sl]] *= fact[sl][:, None]
            L[bk] = L[bk].clone()
            L[bk][cht[sl]] = torch.from_numpy(sgn_o[sl] * fac[sl] / lsdiv)
    for i, L in enumerate(w["layers"]):
        if plant and i > early:
            for k, (a, b) in (("ln1_g", (0.02, 3.0)), ("ln2_g", (3.0, 0.02))):
                L[k] = L[k].clone()
                L[k][cht[0::2]] = a
                L[k][cht[1::2]] = b
        if spec.layerscale:
            for k in ("ls1", "ls2"):
                mag = np.exp(rs.uniform(np.log(ls_range[0]), np.log(ls_range[1]), d)).astype(np.float32)
                sgn = np.where(rs.rand(d) < 0.5, -1.0, 1.0).astype(np.float32)
                v = torch.from_numpy(mag * sgn)
                if plant:
                    v[cht] = 0.5
                L[k] = v
        if sharp and i % 3 == 2:
            L["wqkv"] = L["wqkv"].clone()
            L["bqkv"] = L["bqkv"].clone()
            dh = d // spec.heads
            for hd in (0, spec.heads - 1):
                for base in (0, d):                                  # Q rows, K rows of the head
                    L["wqkv"][base + hd * dh: base + (hd + 1) * dh] *= sharp
                    L["bqkv"][base + hd * dh: base + (hd + 1) * dh] *= sharp
    tok = None
    if profile in ("const", "all"):
        tok = min(int(const_token) + (1 if spec.has_cls else 0), spec.tokens - 1)
        w["pos"] = w["pos"].clone()
        w["pos"][tok] += const_value
    w["hostile"] = {"profile": profile, "channels": ch.tolist() if plant else [], "factors": fac.tolist() if plant else [], "const_token": tok,
                    "early": early, "mid": mid}
    return w
