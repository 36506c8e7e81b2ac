"""DiffVisionTower on MI355X - drop-in for llava/model/multimodal_encoder/diffLVLM/diffusion_encoder.py:15-117.

Same registry tables, constructor arguments (`args.{up_ft_index, t, prompt, vision_tower, ensemble_size, img_size}`),
`DiffImageProcessor.preprocess` and `forward(images)` contract ([B, h*w, C] features).  The SD-UNet featurizer
(SD1.5 / SD2.1 / SDXL), the image-variation featurizer, the DiT featurizer and the SD3 (MMDiT) featurizer all run on the HIP
path.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .src.models.dift_dit import DiTFeaturizer
from .src.models.dift_imsd import IMSDFeaturizer
from .src.models.dift_sd import SDFeaturizer
from .src.models.dift_sd3 import SD3Featurizer


build_featurelizer_mapping = {'lambdalabs/sd-image-variations-diffusers': IMSDFeaturizer,
                              'stabilityai/stable-diffusion-2-1': SDFeaturizer,
                              'runwayml/stable-diffusion-v1-5': SDFeaturizer,
                              'stabilityai/stable-diffusion-xl-base-1.0': SDFeaturizer,
                              'facebook/DiT-XL-2-512': DiTFeaturizer,
                              'stabilityai/stable-diffusion-3-medium-diffusers': SD3Featurizer}

feature_hid_size_mapping = {'runwayml/stable-diffusion-v1-5_feature': 1280,
                            'lambdalabs/sd-image-variations-diffusers': 1280,
                            'runwayml/stable-diffusion-v1-5': 1280,
                            'stabilityai/stable-diffusion-xl-base-1.0': 1280,
                            'stabilityai/stable-diffusion-2-1': 1280,
                            'facebook/DiT-XL-2-512': 4608,
                            'stabilityai/stable-diffusion-3-medium-diffusers': 6144}


class DiffImageProcessor(nn.Module):
    def __init__(self, img_size):
        super().__init__()
        self.img_size = img_size
        self.crop_size = {'height': img_size[0], 'width': img_size[1]}

    def preprocess(self, img, return_tensors: Optional[str] = None, **kwargs):
        if self.img_size[0] > 0:
            img = img.resize(self.img_size)
        arr = np.asarray(img.convert("RGB") if getattr(img, "mode", "RGB") != "RGB" else img)
        img_tensor = (torch.from_numpy(arr.copy()).permute(2, 0, 1) / 255.0 - 0.5) * 2      # PILToTensor()(img) / 255 ...
        return {"pixel_values": [img_tensor]}


class DiffVisionTower(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.is_loaded = False
        self.up_ft_index = args.up_ft_index
        self.t = args.t
        self.prompt = args.prompt
        self.model_id = args.vision_tower
        self.ensemble_size = args.ensemble_size
        self.img_size = [args.img_size, args.img_size]
        self.hidden_size_num = feature_hid_size_mapping[args.vision_tower]
        self.load_model()

    def load_model(self):
        self.image_processor = DiffImageProcessor(self.img_size)
        self.vision_tower = build_featurelizer_mapping[self.model_id](self.model_id)
        self.is_loaded = True

    @torch.no_grad()
    def forward(self, images):
        kw = dict(prompt=self.prompt, t=self.t, up_ft_index=self.up_ft_index, ensemble_size=self.ensemble_size)
        if type(images) is list:
            # the reference appends per-image features to a python list and then calls `.shape` on it (diffusion_encoder.py:71-84),
            # which raises; stacking them is the evident intent
            image_features = torch.stack([self.vision_tower.forward(im if im.dim() == 4 else im.unsqueeze(0), **kw) for im in images])
        else:
            if len(images.shape) == 3:
                images = torch.unsqueeze(images, dim=0)
            image_features = self.vision_tower.forward(images, **kw)
        if len(image_features.shape) == 3:
            image_features = torch.unsqueeze(image_features, dim=0)
        image_features = image_features.permute(0, 2, 3, 1)
        B, H, W, C = image_features.shape
        return image_features.reshape(B, -1, C)

    @property
    def dtype(self):
        return self.vision_tower.dtype

    @property
    def device(self):
        return self.vision_tower.device

    @property
    def hidden_size(self):
        return self.hidden_size_num
