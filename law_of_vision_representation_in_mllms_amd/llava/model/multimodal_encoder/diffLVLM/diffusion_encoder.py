"""DiffVisionTower on MI355X - drop-in for llava/model/multimodal_encoder/diffLVLM/diffusion_encoder.py:15-117.

Same registry tables (`build_featurelizer_mapping`, `feature_hid_size_mapping`), constructor arguments
(`args.{up_ft_index, t, prompt, vision_tower, ensemble_size, img_size}`), `DiffImageProcessor.preprocess` and
`forward(images)` contract ([B, h*w, C] features).  The SD-UNet featurizer (SD1.5 / SD2.1 / SDXL), the image-variation
featurizer, the DiT featurizer and the SD3 (MMDiT) featurizer all run on the HIP path.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .src.models.dift_dit import DiTFeaturizer
from .src.models.dift_imsd import IMSDFeaturizer
from .src.models.dift_sd import SDFeaturizer
from .src.models.dift_sd3 import SD3Featurizer

# tower id -> (featurizer class, channels of the captured feature map); the '_feature' id is a pre-extracted-feature alias that
# only has a width (diffusion_encoder.py:15-28)
_TOWERS = {
    'runwayml/stable-diffusion-v1-5': (SDFeaturizer, 1280),
    'stabilityai/stable-diffusion-2-1': (SDFeaturizer, 1280),
    'stabilityai/stable-diffusion-xl-base-1.0': (SDFeaturizer, 1280),
    'lambdalabs/sd-image-variations-diffusers': (IMSDFeaturizer, 1280),
    'facebook/DiT-XL-2-512': (DiTFeaturizer, 4608),
    'stabilityai/stable-diffusion-3-medium-diffusers': (SD3Featurizer, 6144),
}
build_featurelizer_mapping = {name: cls for name, (cls, _) in _TOWERS.items()}
feature_hid_size_mapping = {**{name: width for name, (_, width) in _TOWERS.items()}, 'runwayml/stable-diffusion-v1-5_feature': 1280}


class DiffImageProcessor(nn.Module):
    """PIL image -> [3, H, W] float in [-1, 1] (resize to img_size when it is positive; diffusion_encoder.py:30-43)."""

    def __init__(self, img_size):
        super().__init__()
        self.img_size = img_size
        self.crop_size = dict(height=img_size[0], width=img_size[1])

    def preprocess(self, img, return_tensors: Optional[str] = None, **kwargs):
        if self.img_size[0] > 0:
            img = img.resize(self.img_size)
        if getattr(img, "mode", "RGB") != "RGB":
            img = img.convert("RGB")
        u8 = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1)            # what PILToTensor() returns
        return {"pixel_values": [(u8 / 255.0 - 0.5) * 2]}

    def device_twin(self, device, dtype=torch.float32):
        """The same preprocessing on the GPU for the device input pipeline (llava/feature/extract.inference(device_decode=True)):
        `img.resize(img_size)` is PIL's default BICUBIC to exactly img_size (no aspect handling, no crop) and (u8 / 255 - 0.5) * 2 equals
        (u8 / 255 - 0.5) / 0.5 bit for bit (a power of two), i.e. a DevicePreprocessor with square_resize, mean = std = 0.5.  Needs a positive
        img_size: without the resize the images of a batch have different sizes."""
        if self.img_size[0] <= 0 or self.img_size[0] != self.img_size[1]:
            raise ValueError("the device input pipeline batches images: DiffImageProcessor needs a positive square img_size")
        from ..... import device_preprocess as DP
        side = int(self.img_size[0])
        return DP.DevicePreprocessor(side, side, [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], square_resize=True, device=device, dtype=dtype)


class DiffVisionTower(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.is_loaded = False
        self.model_id = args.vision_tower
        self.hidden_size_num = feature_hid_size_mapping[self.model_id]             # KeyError for unknown ids, as the reference
        self.t, self.up_ft_index, self.prompt, self.ensemble_size = args.t, args.up_ft_index, args.prompt, args.ensemble_size
        self.img_size = [args.img_size] * 2
        self.load_model()

    def load_model(self):
        self.image_processor = DiffImageProcessor(self.img_size)
        self.vision_tower = build_featurelizer_mapping[self.model_id](self.model_id)
        self.is_loaded = True

    def _featurize(self, px):
        return self.vision_tower.forward(px, prompt=self.prompt, t=self.t, up_ft_index=self.up_ft_index, ensemble_size=self.ensemble_size)

    @torch.no_grad()
    def forward(self, images):
        if isinstance(images, list):
            # the reference appends per-image features to a python list and then calls `.shape` on it (diffusion_encoder.py:71-84),
            # which raises; stacking them is the evident intent
            maps = torch.stack([self._featurize(im if im.dim() == 4 else im[None]) for im in images])
        else:
            maps = self._featurize(images[None] if images.dim() == 3 else images)
        if maps.dim() == 3:                                                        # featurizers squeeze a batch of one
            maps = maps[None]
        B, C = maps.shape[:2]
        return maps.permute(0, 2, 3, 1).reshape(B, -1, C)                          # [B, C, h, w] -> [B, h*w, C]

    @property
    def dtype(self):
        return self.vision_tower.dtype

    @property
    def device(self):
        return self.vision_tower.device

    @property
    def hidden_size(self):
        return self.hidden_size_num
