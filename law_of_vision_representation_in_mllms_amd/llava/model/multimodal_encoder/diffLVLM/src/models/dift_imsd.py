"""IMSDFeaturizer on MI355X - drop-in for diffLVLM/src/models/dift_imsd.py:187-230 (lambdalabs/sd-image-variations-diffusers).

Same UNet / VAE / DDIM path as SDFeaturizer; the cross-attention context is not a text prompt but each image's own CLIP
image embedding: bilinear resize to 224x224 (no CLIP mean/std normalisation - the reference feeds the [-1, 1] tensor as
is, dift_imsd.py:215-216), `CLIPVisionModelWithProjection(...).image_embeds`, `unsqueeze(1)` -> [B, 1, 768].
`prompt` is ignored, as in the reference.
"""
import os

import torch

from law_of_vision_representation_in_mllms_amd import vit_weights as VW
from law_of_vision_representation_in_mllms_amd.image_embed import ClipImageEmbedder, resize_bilinear
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder._vit_tower import _find_local_checkpoint
from law_of_vision_representation_in_mllms_amd.sd_engine import SdEngine
from law_of_vision_representation_in_mllms_amd.sd_weights import SD_SPECS, SdSpec, synthetic_unet, synthetic_vae

from .dift_sd import _load_dir, spec_from_checkpoint

IMSD_ID = 'lambdalabs/sd-image-variations-diffusers'


def synthetic_image_encoder(spec, proj_dim, seed):
    import numpy as np
    w = VW.synthetic_weights(spec, seed=seed)
    rs = np.random.RandomState(seed + 1)
    g = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(spec.d)).astype(np.float32))
    b = torch.from_numpy((0.05 * rs.standard_normal(spec.d)).astype(np.float32))
    p = torch.from_numpy((rs.standard_normal((proj_dim, spec.d)) / np.sqrt(spec.d)).astype(np.float32))
    return w, g, b, p


class IMSDFeaturizer:
    def __init__(self, sd_id=IMSD_ID, device=None, synthetic=None):
        self.sd_id = sd_id
        self.device = torch.device(device if device is not None else "cuda")
        synthetic = os.environ.get("VISREP_SYNTHETIC_WEIGHTS") == "1" if synthetic is None else synthetic
        root = None if synthetic else _find_local_checkpoint(sd_id)
        if root is not None:
            self.spec, _ = spec_from_checkpoint(sd_id, root, need_text=False)
            self._wu, self._wv = _load_dir(os.path.join(root, "unet")), _load_dir(os.path.join(root, "vae"))
            from transformers import CLIPVisionConfig
            enc = os.path.join(root, "image_encoder")
            cfg = CLIPVisionConfig.from_pretrained(enc)
            vspec = VW.spec_from_hf_config(cfg, "imsd-image-encoder")
            sd = _load_dir(enc)
            packed = VW.pack_hf_state_dict(sd, vspec)
            s = {k.replace("vision_model.", "", 1): v for k, v in sd.items()}
            self.embedder = ClipImageEmbedder(vspec, packed, s["post_layernorm.weight"], s["post_layernorm.bias"],
                                              sd["visual_projection.weight"], self.device)
        elif synthetic:
            base = SD_SPECS['runwayml/stable-diffusion-v1-5']                 # the image-variation UNet is the SD1.x architecture
            self.spec = SdSpec(sd_id, base.unet, base.vae, base.sched, text_len=1)
            n_up = len(self.spec.unet.block_out)
            self._wu, self._wv = synthetic_unet(self.spec.unet, 41, n_up_blocks=n_up), synthetic_vae(self.spec.vae, 42)
            vspec = VW.SPECS["openai/clip-vit-large-patch14"]
            w, g, b, p = synthetic_image_encoder(vspec, self.spec.unet.cross_dim, 43)
            self.embedder = ClipImageEmbedder(vspec, w, g, b, p, self.device)
        else:
            raise OSError(f"{sd_id} is not a local diffusers checkpoint directory and is not in the offline HF cache "
                          "(set VISREP_SYNTHETIC_WEIGHTS=1 for deterministic random-init weights)")
        self._engines = {}
        self.dtype = torch.bfloat16

    def _engine(self, up_ft_index) -> SdEngine:
        if up_ft_index not in self._engines:
            self._engines[up_ft_index] = SdEngine(self.spec, self._wu, self._wv, self.device, up_ft_index=up_ft_index)
        return self._engines[up_ft_index]

    def encode_image(self, img_tensor: torch.Tensor) -> torch.Tensor:
        """[B, 3, H, W] in [-1, 1] -> [B, 1, cross_dim] (dift_imsd.py:213-220, pipeline `_encode_image`)."""
        s = self.embedder.spec.image_size
        px = resize_bilinear(img_tensor.to(self.device).float(), (s, s))
        return self.embedder.forward(px).unsqueeze(1)

    @torch.no_grad()
    def forward(self, img_tensor, prompt, t=1, up_ft_index=0, ensemble_size=1, post_noise=None, ddim_noise=None):
        eng = self._engine(up_ft_index)
        B = img_tensor.shape[0]
        ctx = self.encode_image(img_tensor)
        if ensemble_size > 1:
            ctx = ctx.repeat_interleave(ensemble_size, dim=0)                  # same embedding for every ensemble copy
        tokens = eng.forward(img_tensor, None, t=t, ensemble_size=ensemble_size, post_noise=post_noise, ddim_noise=ddim_noise,
                             image_context=ctx)
        f = 2 ** (len(self.spec.vae.block_out) - 1)
        lh, lw = img_tensor.shape[2] // f, img_tensor.shape[3] // f
        h = int(round((tokens.shape[1] * lh / lw) ** 0.5))
        w = tokens.shape[1] // h
        return tokens.view(B, 1, h, w, tokens.shape[2]).permute(0, 1, 4, 2, 3).squeeze()
