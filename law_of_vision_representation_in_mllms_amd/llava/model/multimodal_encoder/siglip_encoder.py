"""SigLipVisionTower on MI355X — drop-in for llava/model/multimodal_encoder/siglip_encoder.py:7-79.

The reference hard-codes select_feature='cls_patch' (keep all 196 tokens, l.15), dtype float16 / device "cuda" /
hidden_size 768 (l.58-75); the first is kept, the properties report the real engine values (bf16, the tower's device).
"""
from ._vit_tower import HipViTTower


class SigLipVisionTower(HipViTTower):
    FAMILY = "siglip"
    DEFAULT_SELECT_FEATURE = "cls_patch"

    def __init__(self, vision_tower, args, delay_load=False):
        super().__init__(vision_tower, args, delay_load=delay_load)
        self.select_feature = "cls_patch"
