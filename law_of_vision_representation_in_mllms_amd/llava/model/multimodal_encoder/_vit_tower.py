"""Common base of the MI355X ViT vision towers (host side).

Mirrors the tower protocol of the reference plug-ins (llava/model/multimodal_encoder/clip_encoder.py:7-78):
ctor (vision_tower, args, delay_load=False), load_model(), feature_select(), forward(images), attributes is_loaded /
vision_tower_name / select_layer / select_feature / image_processor / vision_tower, properties dummy_feature, dtype,
device, config, hidden_size, num_patches.  The arithmetic runs in libvisrep_hip.so through engine.VitEngine; HuggingFace
is used only to READ checkpoints/configs when they exist locally.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from .... import vit_weights as VW


class _HiddenStates:
    """Lazy stand-in for HF's `output_hidden_states` tuple: index k runs the HIP tower for k layers."""

    def __init__(self, engine, pixels):
        self._e, self._px = engine, pixels
        self._n = engine.n_layers + 1

    def __len__(self):
        return self._n

    def __getitem__(self, k):
        idx = k if k >= 0 else self._n + k
        if not 0 <= idx < self._n:
            raise IndexError("tuple index out of range")
        return self._e.forward(self._px, n_layers=idx)


class _EngineModule(nn.Module):
    """`self.vision_tower` of the reference towers: callable like the HF model, exposes .dtype/.device/.config."""

    def __init__(self, engine, config, dtype=torch.bfloat16):
        super().__init__()
        self.engine = engine
        self.config = config
        self._dtype = dtype
        self._anchor = nn.Parameter(torch.zeros(1, dtype=dtype, device=engine.device), requires_grad=False)

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self.engine.device

    def forward(self, pixel_values, output_hidden_states=True):
        return SimpleNamespace(hidden_states=_HiddenStates(self.engine, pixel_values))


def _find_local_checkpoint(name: str) -> Optional[str]:
    """A local checkpoint directory - transformers layout (config.json) or diffusers layout (model_index.json: unet/ vae/ ... below it) -
    else the offline HF cache's snapshot of `name`, else None."""
    if os.path.isdir(name) and any(os.path.exists(os.path.join(name, f)) for f in ("config.json", "model_index.json")):
        return name
    try:                                                     # offline HF cache
        from huggingface_hub import snapshot_download
        return snapshot_download(name, local_files_only=True)
    except Exception:
        return None


def _processor_crop(path: str) -> Optional[int]:
    """crop_size of the checkpoint's preprocessor_config.json (the input size the reference's AutoImageProcessor produces)."""
    import json
    f = os.path.join(path, "preprocessor_config.json")
    if not os.path.exists(f):
        return None
    try:
        with open(f) as fh:
            c = json.load(fh).get("crop_size")
    except (OSError, ValueError):
        return None
    if isinstance(c, dict):
        c = c.get("height") or c.get("shortest_edge")
    return int(c) if c else None


def _load_state_dict(path: str):
    import glob
    sd = {}
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        from safetensors.torch import load_file
        for f in st:
            sd.update(load_file(f))
        return sd
    for f in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
        sd.update(torch.load(f, map_location="cpu", weights_only=True))
    if not sd:
        raise OSError(f"no weights found under {path}")
    return sd


class HipViTTower(nn.Module):
    FAMILY = "clip"
    DEFAULT_SELECT_FEATURE = "patch"

    def __init__(self, vision_tower, args, delay_load=False):
        super().__init__()
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = args.mm_vision_select_layer
        self.select_feature = getattr(args, "mm_vision_select_feature", self.DEFAULT_SELECT_FEATURE)
        # C-score path runs DINOv2 at 224 or 336.  NOT `args.img_size`: that is the diffusion towers' field, and LLaVA's
        # ModelArguments always carries it (default 768, train.py:87) — the reference's ViT towers ignore it
        self._img_size = getattr(args, "vit_img_size", None)
        self._synthetic = bool(getattr(args, "synthetic_weights", False)) or os.environ.get("VISREP_SYNTHETIC_WEIGHTS") == "1"
        self._device = getattr(args, "device", None)
        # 'bf16' (default; what LLaVA runs the tower in: model.to(bfloat16)) or 'fp32' (what C_score/extract_feature.py runs the
        # CLIP / OpenCLIP / DINOv2 towers in: no dtype cast, fp32 pixels) - VISREP_TOWER_PRECISION overrides
        self._precision = os.environ.get("VISREP_TOWER_PRECISION") or getattr(args, "tower_precision", None) or "bf16"
        # fp32 towers only: the split-bf16 product set (engine.VitEngineF32); None = the fp32-equivalent default (6).  3 is an explicit
        # opt-in for throughput runs (the sweep), never implied
        self._products = getattr(args, "tower_products", None)
        if not delay_load:
            self.load_model()
        else:
            self.cfg_only = self._config_only()

    # ------------------------------------------------------------------ loading
    def _spec_and_weights(self):
        path = _find_local_checkpoint(self.vision_tower_name)
        if path is not None:
            from transformers import AutoConfig
            cfg = AutoConfig.from_pretrained(path)
            spec = VW.spec_from_hf_config(cfg, self.vision_tower_name, crop_size=_processor_crop(path))
            sd = _load_state_dict(path)
            sd = {k: v for k, v in sd.items() if not k.startswith(("text_model.", "logit_", "text_projection", "visual_projection"))}
            return spec, VW.pack_hf_state_dict(sd, spec)
        if self._synthetic and self.vision_tower_name in VW.SPECS:
            spec = VW.SPECS[self.vision_tower_name]
            native = spec.pos_grid * spec.patch if spec.pos_grid else spec.image_size
            base = spec.at_resolution(native)
            w = VW.synthetic_weights(base, seed=1)
            return VW.weights_at_resolution(base, w, spec.image_size)
        raise OSError(f"{self.vision_tower_name} is not a local checkpoint directory and is not in the offline HF cache "
                      "(set VISREP_SYNTHETIC_WEIGHTS=1 or args.synthetic_weights for deterministic random-init weights)")

    def _config_only(self):
        path = _find_local_checkpoint(self.vision_tower_name)
        if path is not None:
            from transformers import AutoConfig
            return AutoConfig.from_pretrained(path)
        if self.vision_tower_name not in VW.SPECS:           # like from_pretrained on a name that is neither a directory nor cached
            raise OSError(f"{self.vision_tower_name} is not a local checkpoint directory, is not in the offline HF cache and has no built-in "
                          "architecture")
        spec = VW.SPECS[self.vision_tower_name]
        return SimpleNamespace(hidden_size=spec.d, image_size=spec.image_size, patch_size=spec.patch,
                               num_hidden_layers=spec.layers, num_attention_heads=spec.heads, intermediate_size=spec.mlp)

    def _make_image_processor(self, spec):
        from .image_processing import default_image_processor
        return default_image_processor(spec)

    def load_model(self):
        from .... import engine
        spec, w = self._spec_and_weights()
        if self._img_size and self._img_size != spec.image_size:
            spec, w = VW.weights_at_resolution(spec, w, self._img_size)
        self.spec = spec
        self.image_processor = self._make_image_processor(spec)
        cfg = SimpleNamespace(hidden_size=spec.d, image_size=spec.image_size, patch_size=spec.patch,
                              num_hidden_layers=spec.layers, num_attention_heads=spec.heads, intermediate_size=spec.mlp)
        fp32 = self._precision in ("fp32", "float32")
        eng = engine.make_engine(spec, w, self._device, "fp32" if fp32 else "bf16", products=self._products if fp32 else None)
        if isinstance(eng, engine.VitEngineCPU):                            # args.device = "cpu": the host twin (explicit opt-in; BASELINE configs[0])
            fp32 = True
        self.vision_tower = _EngineModule(eng, cfg, torch.float32 if fp32 else torch.bfloat16)
        self.vision_tower.requires_grad_(False)
        self.is_loaded = True

    # ------------------------------------------------------------------ reference protocol
    def feature_select(self, image_forward_outs):
        image_features = image_forward_outs.hidden_states[self.select_layer]
        if self.select_feature == "patch":
            image_features = image_features[:, 1:]
        elif self.select_feature == "cls_patch":
            image_features = image_features
        else:
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        return image_features

    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:
            image_features = []
            for image in images:
                out = self.vision_tower(image.to(device=self.device).unsqueeze(0), output_hidden_states=True)
                image_features.append(self.feature_select(out).to(image.dtype))
        else:
            out = self.vision_tower(images.to(device=self.device), output_hidden_states=True)
            image_features = self.feature_select(out).to(images.dtype)
        return image_features

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.vision_tower.dtype

    @property
    def device(self):
        return self.vision_tower.device

    @property
    def config(self):
        return self.vision_tower.config if self.is_loaded else self.cfg_only

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2
