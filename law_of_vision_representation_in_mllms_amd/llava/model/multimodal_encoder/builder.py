"""Tower builders of the MI355X towers.

Public names, argument meaning and error behaviour follow llava/model/multimodal_encoder/builder.py:8-31 (the registry in
llava_arch.py maps tower ids onto these five functions): the tower id is `cfg.mm_vision_tower`, else `cfg.vision_tower`; the
CLIP builder accepts an existing local path or an id starting with "openai" / "laion" and raises
ValueError("Unknown vision tower: ...") otherwise; the diffusion builder reads everything from the config object; 'feature'
towers are pre-extracted features and have no module.
"""
import os

from .clip_encoder import CLIPVisionTower
from .diffLVLM.diffusion_encoder import DiffVisionTower
from .dinov2_encoder import DinoV2VisionTower
from .siglip_encoder import SigLipVisionTower

_CLIP_PREFIXES = ("openai", "laion")


def _tower_id(cfg):
    name = getattr(cfg, 'mm_vision_tower', None)
    return name if name is not None else getattr(cfg, 'vision_tower', None)


def _make(cls, cfg, kwargs):
    return cls(_tower_id(cfg), args=cfg, **kwargs)


def build_vision_tower(vision_tower_cfg, **kwargs):
    name = _tower_id(vision_tower_cfg)
    if not (os.path.exists(name) or name.startswith(_CLIP_PREFIXES)):
        raise ValueError(f'Unknown vision tower: {name}')
    return _make(CLIPVisionTower, vision_tower_cfg, kwargs)


def build_dinov2_vision_tower(vision_tower_cfg, **kwargs):
    return _make(DinoV2VisionTower, vision_tower_cfg, kwargs)


def build_siglip_vision_tower(vision_tower_cfg, **kwargs):
    return _make(SigLipVisionTower, vision_tower_cfg, kwargs)


def build_diffusion_vision_tower(vision_tower_cfg, **kwargs):
    """SD1.5 / SD2.1 / SDXL / image-variations UNets, DiT-XL/2 and SD3-medium featurizers — all on the HIP path."""
    return DiffVisionTower(args=vision_tower_cfg)


def build_feature(vision_tower_cfg, **kwargs):
    return 'feature'
