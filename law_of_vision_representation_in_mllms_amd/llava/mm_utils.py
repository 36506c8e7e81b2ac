"""Host-side image pre-processing — same functions as llava/mm_utils.py:64-95."""
import torch
from PIL import Image


def expand2square(pil_img, background_color):
    width, height = pil_img.size
    if width == height:
        return pil_img
    side = max(width, height)
    result = Image.new(pil_img.mode, (side, side), background_color)
    if width > height:
        result.paste(pil_img, (0, (width - height) // 2))
    else:
        result.paste(pil_img, ((height - width) // 2, 0))
    return result


def process_images(images, image_processor, model_cfg):
    image_aspect_ratio = getattr(model_cfg, "image_aspect_ratio", None)
    new_images = []
    if image_aspect_ratio == 'pad':
        for image in images:
            image = expand2square(image, tuple(int(x * 255) for x in image_processor.image_mean))
            new_images.append(image_processor.preprocess(image, return_tensors='pt')['pixel_values'][0])
    else:
        if type(image_processor) is list:
            return [p.preprocess(images[0], return_tensors='pt')['pixel_values'] for p in image_processor]
        return image_processor.preprocess(images[0], return_tensors='pt')['pixel_values'][0]
    if all(x.shape == new_images[0].shape for x in new_images):
        new_images = torch.stack(new_images, dim=0)
    return new_images
