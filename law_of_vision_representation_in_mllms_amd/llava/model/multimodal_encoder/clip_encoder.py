"""CLIPVisionTower on MI355X — drop-in for llava/model/multimodal_encoder/clip_encoder.py:7-78 (CLIP and OpenCLIP ids)."""
from ._vit_tower import HipViTTower


class CLIPVisionTower(HipViTTower):
    FAMILY = "clip"
    DEFAULT_SELECT_FEATURE = "patch"
