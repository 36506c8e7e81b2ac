"""SD3Featurizer on MI355X - drop-in for diffLVLM/src/models/dift_sd3.py:122-175 (stabilityai/stable-diffusion-3-medium-diffusers).

Same constructor and `forward(img_tensor, prompt, t=1, up_ft_index=-1, ensemble_size=1)` signature.  As in the reference the T5
encoder is not loaded (`text_encoder_3=None`, dift_sd3.py:127-128): the prompt is CLIP-L | CLIP-G `hidden_states[-2]`
zero-padded to 4096 columns, followed by 256 zero rows standing in for T5; the pooled projection is the two encoders'
`text_embeds` concatenated (vendored pipeline_stable_diffusion_3.py `_get_clip_prompt_embeds` / `encode_prompt`).
The reference's forward ends without `return` (dift_sd3.py:175); this class returns the 2x2-unfolded block output it computes.
"""
import os

import torch

from law_of_vision_representation_in_mllms_amd.engine import gemm
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder._vit_tower import _find_local_checkpoint
from law_of_vision_representation_in_mllms_amd.sd3_engine import Sd3Engine
from law_of_vision_representation_in_mllms_amd.sd_weights import (SD3_SPECS, Sd3CoreSpec, Sd3Spec, TextSpec, VaeSpec, _synthetic, synthetic_sd3,
                                                                   synthetic_text, synthetic_vae)
from law_of_vision_representation_in_mllms_amd.text_engine import ClipTextEngine

from .dift_sd import _json, _load_dir

T5_ROWS = 256          # max_sequence_length of the absent T5 branch: zeros [1, 256, joint_dim] (pipeline `_get_t5_prompt_embeds`)
_SYNTH_TEXT = (TextSpec(), TextSpec(d=1280, mlp=5120, layers=32, heads=20, act="gelu"))


def spec_from_checkpoint(name, root) -> Sd3Spec:
    t = _json(os.path.join(root, "transformer", "config.json"))
    v = _json(os.path.join(root, "vae", "config.json"))
    core = Sd3CoreSpec(heads=t["num_attention_heads"], head_dim=t["attention_head_dim"], in_channels=t["in_channels"], layers=t["num_layers"],
                       sample_size=t["sample_size"], patch=t["patch_size"], joint_dim=t["joint_attention_dim"],
                       pooled_dim=t["pooled_projection_dim"], pos_max=t["pos_embed_max_size"])
    vae = VaeSpec(in_channels=v["in_channels"], block_out=tuple(v["block_out_channels"]), layers_per_block=v["layers_per_block"],
                  latent_channels=v["latent_channels"], groups=v["norm_num_groups"], scaling_factor=v.get("scaling_factor", 1.5305),
                  quant_conv=bool(v.get("use_quant_conv", True)))
    return Sd3Spec(name, core, vae)


def _text_spec(cfg) -> TextSpec:
    return TextSpec(vocab=cfg["vocab_size"], d=cfg["hidden_size"], mlp=cfg["intermediate_size"], layers=cfg["num_hidden_layers"],
                    heads=cfg["num_attention_heads"], max_pos=cfg["max_position_embeddings"], act=cfg.get("hidden_act", "quick_gelu"),
                    eps=cfg.get("layer_norm_eps", 1e-5))


class SD3Featurizer:
    def __init__(self, sd_id="stabilityai/stable-diffusion-3-medium-diffusers", device=None, synthetic=None):
        self.sd_id = sd_id
        self.device = torch.device(device if device is not None else "cuda")
        synthetic = os.environ.get("VISREP_SYNTHETIC_WEIGHTS") == "1" if synthetic is None else synthetic
        root = None if synthetic else _find_local_checkpoint(sd_id)
        self.tokenizers = [None, None]
        self.text, self.text_proj = [], []
        if root is not None:
            self.spec = spec_from_checkpoint(sd_id, root)
            self._wc, self._wv = _load_dir(os.path.join(root, "transformer")), _load_dir(os.path.join(root, "vae"))
            from transformers import CLIPTokenizer
            for i, sub in enumerate(("text_encoder", "text_encoder_2")):
                sd = _load_dir(os.path.join(root, sub))
                wt = {k.replace("text_model.", "", 1): v for k, v in sd.items()}
                self.text.append(ClipTextEngine(_text_spec(_json(os.path.join(root, sub, "config.json"))), wt, self.device))
                self.text_proj.append(sd["text_projection.weight"])
                self.tokenizers[i] = CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer" if i == 0 else "tokenizer_2"))
        elif synthetic:
            self.spec = SD3_SPECS[sd_id]
            self._wc, self._wv = synthetic_sd3(self.spec.core, 61), synthetic_vae(self.spec.vae, 62)
            for i, ts in enumerate(_SYNTH_TEXT):
                self.text.append(ClipTextEngine(ts, synthetic_text(ts, 63 + i), self.device))
                self.text_proj.append(_synthetic([("text_projection.weight", (ts.d, ts.d))], 65 + i)["text_projection.weight"])
        else:
            raise OSError(f"{sd_id} is not a local diffusers checkpoint directory and is not in the offline HF cache "
                          "(set VISREP_SYNTHETIC_WEIGHTS=1 for deterministic random-init weights)")
        self.text_proj = [p.to(self.device, torch.bfloat16).contiguous() for p in self.text_proj]
        self._engines = {}
        self._prompt_cache = {}
        self.dtype = torch.bfloat16

    def _engine(self, up_ft_index) -> Sd3Engine:
        if up_ft_index not in self._engines:
            self._engines[up_ft_index] = Sd3Engine(self.spec, self._wc, self._wv, self.device, up_ft_index=up_ft_index)
        return self._engines[up_ft_index]

    def tokenize(self, prompt: str, which: int) -> torch.Tensor:
        ts = self.text[which].spec
        tok = self.tokenizers[which]
        if tok is not None:
            return tok(prompt, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        v, L = ts.vocab, min(77, ts.max_pos)                  # synthetic weights: byte-level stand-in, <bos> bytes <eos>-padding
        body = [b % (v - 2) for b in prompt.encode("utf-8")][: L - 2]
        return torch.tensor([[v - 2] + body + [v - 1] * (L - 1 - len(body))], dtype=torch.long)

    def encode_prompt(self, prompt: str):
        """(prompt_embeds [1, 77 + 256, joint_dim], pooled [1, pooled_dim]) as pipe.encode_prompt(prompt, None, None) gives
        with text_encoder_3 = None (dift_sd3.py:152-158)."""
        if prompt not in self._prompt_cache:
            c = self.spec.core
            embeds, pooled = [], []
            for i, eng in enumerate(self.text):
                ids = self.tokenize(prompt, i)
                embeds.append(eng.forward(ids, hidden_state=-2))                               # hidden_states[-2]
                last = eng.forward(ids)                                                        # after final_layer_norm
                eos = int(ids[0].argmax())                                                     # CLIPTextTransformer pooled_output
                pooled.append(gemm(last[0, eos: eos + 1].contiguous(), self.text_proj[i]))     # text_embeds = text_projection(pooled)
            clip = torch.cat(embeds, dim=-1)                                                   # [1, 77, 768 + 1280]
            pe = torch.zeros(1, clip.shape[1] + T5_ROWS, c.joint_dim, dtype=torch.bfloat16, device=self.device)
            pe[:, : clip.shape[1], : clip.shape[2]] = clip                                     # F.pad to joint_dim, then the T5 zeros
            self._prompt_cache[prompt] = (pe, torch.cat(pooled, dim=-1).float())
        return self._prompt_cache[prompt]

    @torch.no_grad()
    def forward(self, img_tensor, prompt, t=1, up_ft_index=-1, ensemble_size=1, post_noise=None, ddim_noise=None):
        pe, pooled = self.encode_prompt(prompt)
        tokens = self._engine(up_ft_index).forward(img_tensor, pe, t=t, ensemble_size=ensemble_size, post_noise=post_noise,
                                                   ddim_noise=ddim_noise, pooled=pooled)          # [B, (h/2)(w/2), 4D]
        B, n, C = tokens.shape
        s = int(round(n ** 0.5))
        return tokens.view(B, s, s, C).permute(0, 3, 1, 2)                                        # [B, 4D, h/2, w/2] (a view)
