"""Host-side image pre-processing with the call signatures of llava/mm_utils.py:64-95 (`expand2square`, `process_images`).

Behaviour kept from the reference: 'pad' mode squares every image on a mean-colour canvas (centred along the short side) and
pre-processes all of them, stacking when the shapes agree; every other mode pre-processes ONLY `images[0]` - with a list of
processors ('.'-fused towers) one `[1, 3, H, W]` batch per processor, otherwise the bare `[3, H, W]` tensor.
"""
import torch
from PIL import Image


def expand2square(pil_img, background_color):
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))        # centred on the short axis, flush on the long one
    return canvas


def _pixels(processor, image):
    return processor.preprocess(image, return_tensors='pt')['pixel_values']


def process_images(images, image_processor, model_cfg):
    if getattr(model_cfg, "image_aspect_ratio", None) != 'pad':
        if type(image_processor) is list:
            return [_pixels(p, images[0]) for p in image_processor]
        return _pixels(image_processor, images[0])[0]
    fill = tuple(int(c * 255) for c in image_processor.image_mean)
    out = [_pixels(image_processor, expand2square(im, fill))[0] for im in images]
    same = all(t.shape == out[0].shape for t in out)
    return torch.stack(out, dim=0) if same else out
