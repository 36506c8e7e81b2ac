"""Offline-constructible image processors with the public defaults of each backbone (the reference pulls them from the
hub: clip_encoder.py:23, dinov2_encoder.py:24, siglip_encoder.py:23-24).  Host side, PIL/numpy only."""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def shortest_edge_size(w: int, h: int, size: int):
    """HF get_resize_output_image_size(default_to_square=False): short side -> size, long side -> int(size * long / short)
    (truncation, not rounding: 500 x 375 -> 298 x 224, not 299)."""
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    return (size, new_long) if w <= h else (new_long, size)


class SimpleImageProcessor:
    """resize (shortest edge, bicubic) -> center crop -> /255 -> normalise; `.preprocess(img, return_tensors='pt')`."""

    def __init__(self, resize_to, crop, mean, std, square_resize=False):
        self.resize_to, self.crop = resize_to, crop
        self.image_mean, self.image_std = list(mean), list(std)
        self.crop_size = {"height": crop, "width": crop}
        self.size = {"shortest_edge": resize_to}
        self.square_resize = square_resize

    def _one(self, img: Image.Image) -> torch.Tensor:
        img = img.convert("RGB")
        if self.square_resize:
            img = img.resize((self.crop, self.crop), Image.BICUBIC)
        else:
            w, h = img.size
            nw, nh = shortest_edge_size(w, h, self.resize_to)
            img = img.resize((nw, nh), Image.BICUBIC)
            w, h = img.size
            l, t = (w - self.crop) // 2, (h - self.crop) // 2
            img = img.crop((l, t, l + self.crop, t + self.crop))
        a = np.asarray(img, dtype=np.float32) / 255.0
        a = (a - np.asarray(self.image_mean, np.float32)) / np.asarray(self.image_std, np.float32)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))   # numpy copy: a torch op here would wake the intra-op pool per image

    def preprocess(self, images, return_tensors="pt"):
        if not isinstance(images, (list, tuple)):
            images = [images]
        return {"pixel_values": torch.stack([self._one(i) for i in images])}

    __call__ = preprocess


def default_image_processor(spec):
    if spec.family == "clip":
        return SimpleImageProcessor(spec.image_size, spec.image_size, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD)
    if spec.family == "dinov2":
        return SimpleImageProcessor(int(round(spec.image_size * 256 / 224)), spec.image_size, IMAGENET_MEAN, IMAGENET_STD)
    return SimpleImageProcessor(spec.image_size, spec.image_size, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), square_resize=True)
