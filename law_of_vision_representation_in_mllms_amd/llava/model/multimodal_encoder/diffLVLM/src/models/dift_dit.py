"""DiTFeaturizer on MI355X - drop-in for diffLVLM/src/models/dift_dit.py:158-196 (facebook/DiT-XL-2-512 block features).

Same constructor and `forward(img_tensor, prompt, t=1, up_ft_index=-1, ensemble_size=1)` signature.  The reference's
forward computes the 2x2-unfolded block output [B, 4*D, h/2, w/2] and then falls off the end without returning it
(dift_dit.py:196), so `DiffVisionTower.forward` cannot work with it as committed; this class returns that tensor.
`prompt` is unused by the reference (class-unconditional, timestep-only conditioning) and here.
"""
import os

import torch

from law_of_vision_representation_in_mllms_amd.dit_engine import DiTEngine
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder._vit_tower import _find_local_checkpoint
from law_of_vision_representation_in_mllms_amd.sd_weights import (DIT_SPECS, DiTCoreSpec, DiTSpec, SchedulerSpec, VaeSpec, synthetic_dit,
                                                                   synthetic_vae)

from .dift_sd import _json, _load_dir


def spec_from_checkpoint(name, root) -> DiTSpec:
    t = _json(os.path.join(root, "transformer", "config.json"))
    v = _json(os.path.join(root, "vae", "config.json"))
    s = _json(os.path.join(root, "scheduler", "scheduler_config.json"))
    core = DiTCoreSpec(heads=t["num_attention_heads"], head_dim=t["attention_head_dim"], in_channels=t["in_channels"],
                       layers=t["num_layers"], sample_size=t["sample_size"], patch=t["patch_size"], eps=t.get("norm_eps", 1e-5),
                       num_classes=t.get("num_embeds_ada_norm", 1000))
    vae = VaeSpec(in_channels=v["in_channels"], block_out=tuple(v["block_out_channels"]), layers_per_block=v["layers_per_block"],
                  latent_channels=v["latent_channels"], groups=v["norm_num_groups"], scaling_factor=v.get("scaling_factor", 0.18215))
    sched = SchedulerSpec(num_train_timesteps=s["num_train_timesteps"], beta_start=s["beta_start"], beta_end=s["beta_end"],
                          beta_schedule=s["beta_schedule"])
    return DiTSpec(name, core, vae, sched)


class DiTFeaturizer:
    def __init__(self, sd_id='facebook/DiT-XL-2-512', device=None, synthetic=None):
        self.sd_id = sd_id
        self.device = torch.device(device if device is not None else "cuda")
        synthetic = os.environ.get("VISREP_SYNTHETIC_WEIGHTS") == "1" if synthetic is None else synthetic
        root = None if synthetic else _find_local_checkpoint(sd_id)
        if root is not None:
            self.spec = spec_from_checkpoint(sd_id, root)
            self._wd, self._wv = _load_dir(os.path.join(root, "transformer")), _load_dir(os.path.join(root, "vae"))
        elif synthetic:
            self.spec = DIT_SPECS[sd_id]
            self._wd, self._wv = synthetic_dit(self.spec.core, 31), synthetic_vae(self.spec.vae, 32)
        else:
            raise OSError(f"{sd_id} is not a local diffusers checkpoint directory and is not in the offline HF cache "
                          "(set VISREP_SYNTHETIC_WEIGHTS=1 for deterministic random-init weights)")
        self._engines = {}
        self.dtype = torch.bfloat16

    def _engine(self, up_ft_index) -> DiTEngine:
        if up_ft_index not in self._engines:
            self._engines[up_ft_index] = DiTEngine(self.spec, self._wd, self._wv, self.device, up_ft_index=up_ft_index)
        return self._engines[up_ft_index]

    @torch.no_grad()
    def forward(self, img_tensor, prompt, t=1, up_ft_index=-1, ensemble_size=1, post_noise=None, ddim_noise=None):
        tokens = self._engine(up_ft_index).forward(img_tensor, None, t=t, ensemble_size=ensemble_size, post_noise=post_noise,
                                                   ddim_noise=ddim_noise)                          # [B, (h/2)(w/2), 4D]
        B, n, C = tokens.shape
        s = int(round(n ** 0.5))
        return tokens.view(B, s, s, C).permute(0, 3, 1, 2)                                         # [B, 4D, h/2, w/2] (a view)
