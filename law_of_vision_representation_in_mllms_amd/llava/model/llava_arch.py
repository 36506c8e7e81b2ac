"""Vision side of llava_arch on MI355X: the tower registry, tower(+fusion)/projector wiring, encode_images and the
A-feature dump hook.  Mirrors llava/model/llava_arch.py:29-40 (build_function_mapping), :46-197 (LlavaMetaModel vision
modules incl. '.'-fusion), :229-248 (save_tensor_to_folder), :260-286 (encode_images).  The LLM-side token splicing
(prepare_inputs_labels_for_multimodal) is out of scope.
"""
import os

import torch
import torch.nn as nn

from .multimodal_encoder.builder import (build_diffusion_vision_tower, build_dinov2_vision_tower, build_feature,
                                         build_siglip_vision_tower, build_vision_tower)
from .multimodal_projector.builder import build_vision_projector

build_function_mapping = {'openai/clip-vit-large-patch14-336': build_vision_tower,
                          'google/siglip-base-patch16-224': build_siglip_vision_tower,
                          'laion/CLIP-ViT-L-14-laion2B-s32B-b82K': build_vision_tower,
                          'stabilityai/stable-diffusion-2-1': build_diffusion_vision_tower,
                          'runwayml/stable-diffusion-v1-5': build_diffusion_vision_tower,
                          'lambdalabs/sd-image-variations-diffusers': build_diffusion_vision_tower,
                          'facebook/dinov2-large': build_dinov2_vision_tower,
                          'stabilityai/stable-diffusion-xl-base-1.0': build_diffusion_vision_tower,
                          'feature': build_feature,
                          'facebook/DiT-XL-2-512': build_diffusion_vision_tower,
                          'stabilityai/stable-diffusion-3-medium-diffusers': build_diffusion_vision_tower,
                          'openai/clip-vit-large-patch14': build_vision_tower}


def save_tensor_to_folder(tensor, folder_path, max_tensors=100, exit_when_full=True):
    """llava_arch.py:229-248: dump `tensor_{k}.pt` (k = 1..max_tensors); the reference exit()s when full."""
    if not os.path.exists(folder_path):
        os.makedirs(folder_path)
    tensor_count = len([f for f in os.listdir(folder_path) if f.endswith('.pt')])
    if tensor_count < max_tensors:
        tensor_filename = os.path.join(folder_path, f'tensor_{tensor_count + 1}.pt')
        torch.save(tensor, tensor_filename)
        print(f'Saved tensor to {tensor_filename}')
    if tensor_count + 1 >= max_tensors:
        print(f'Tensor count has reached {max_tensors}. Exiting the program.')
        if exit_when_full:
            exit()
        return True
    return False


class VisionEncoderStack(nn.Module):
    """Vision tower(s) + mm_projector exactly as LlavaMetaModel wires them (llava_arch.py:46-110, 114-197).

    config.mm_vision_tower may be one registry id or several joined with '.' (channel-concat fusion before ONE projector,
    llava_arch.py:70-84,164-168,278-285; e.g. 'openai/clip-vit-large-patch14.facebook/dinov2-large').
    """

    def __init__(self, config, delay_load=False):
        super().__init__()
        self.config = config
        names = self._split(config.mm_vision_tower)
        towers = []
        for n in names:
            if n not in build_function_mapping:
                raise KeyError(n)
            cfg = _CfgView(config, n)
            towers.append(build_function_mapping[n](cfg, delay_load=delay_load))
        self.vision_tower = towers[0] if len(towers) == 1 else nn.ModuleList(towers)
        config.mm_hidden_size = sum(t.hidden_size for t in towers)
        self.mm_projector = build_vision_projector(config)

    @staticmethod
    def _split(spec):
        # registry ids contain '.' themselves ('sd1.5'-style ids do, CLIP/DINOv2 ids do not): greedy match on known ids
        out, rest = [], spec
        keys = sorted(build_function_mapping, key=len, reverse=True)
        while rest:
            for k in keys:
                if rest == k or rest.startswith(k + '.'):
                    out.append(k)
                    rest = rest[len(k) + 1:]
                    break
            else:
                raise KeyError(spec)
        return out

    def get_vision_tower(self):
        return self.vision_tower

    def load_projector(self, path):
        """llava_arch.py:183-189: `pretrain_mm_mlp_adapter` file with keys model.mm_projector.{0,2}.{weight,bias}."""
        weights = torch.load(path, map_location='cpu')
        get_w = lambda w, kw: {k.split(kw + '.')[1]: v for k, v in w.items() if kw in k}
        self.mm_projector.load_state_dict(get_w(weights, 'mm_projector'))

    @torch.no_grad()
    def encode_images(self, images):
        """llava_arch.py:260-286."""
        if type(images) is not list:
            return self.mm_projector(self.vision_tower(images))
        if type(self.vision_tower) is nn.ModuleList:
            f_list = [v(images[i]) for i, v in enumerate(self.vision_tower)]
            return self.mm_projector(torch.cat(f_list, dim=-1))
        raise ValueError("a list of image tensors needs a '.'-fused tower list")

    def encode_features(self, images):
        return self.mm_projector(images)


class _CfgView:
    """Per-tower view of the model config: same attributes, `mm_vision_tower` / `vision_tower` narrowed to one id
    (the reference rewrites config.mm_vision_tower - and config.vision_tower for the diffusion towers - per tower,
    llava_arch.py:61-84; DiffVisionTower reads args.vision_tower, diffusion_encoder.py:52)."""

    def __init__(self, base, name):
        object.__setattr__(self, "_b", base)
        object.__setattr__(self, "mm_vision_tower", name)
        object.__setattr__(self, "vision_tower", name)

    def __getattr__(self, k):
        return getattr(object.__getattribute__(self, "_b"), k)
