"""DinoV2VisionTower on MI355X — drop-in for llava/model/multimodal_encoder/dinov2_encoder.py:8-83."""
from ._vit_tower import HipViTTower


class DinoV2VisionTower(HipViTTower):
    FAMILY = "dinov2"
    DEFAULT_SELECT_FEATURE = "patch"

    def __init__(self, vision_tower, args, delay_load=False):
        super().__init__(vision_tower, args, delay_load=delay_load)
        if delay_load:
            return
        self.cfg_only = self.config                  # the reference always sets cfg_only (dinov2_encoder.py:20-21)
