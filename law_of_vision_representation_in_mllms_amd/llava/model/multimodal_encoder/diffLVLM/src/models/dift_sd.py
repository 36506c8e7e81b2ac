"""SDFeaturizer on MI355X - drop-in for diffLVLM/src/models/dift_sd.py:224-276 (SD1.5 / SD2.1 UNet features).

Same constructor and `forward(img_tensor, prompt, t=1, up_ft_index=0, ensemble_size=1)` contract; underneath, the VAE
encoder, the DDIM noising, the truncated UNet and the CLIP prompt encoder run in libvisrep_hip.so
(sd_engine.SdEngine / text_engine.ClipTextEngine).  Differences that matter to a caller:
  * the two randn draws may be passed in (`post_noise=`, `ddim_noise=`) for reproducible features; omitted, they are drawn
    on the device like the reference does,
  * checkpoints are read from a local diffusers directory or the offline HF cache (unet/, vae/, scheduler/,
    text_encoder/, tokenizer/); with VISREP_SYNTHETIC_WEIGHTS=1 a deterministic random-init model of the same
    architecture is used and prompts are tokenised by bytes (no vocabulary files exist offline),
  * SDXL (`OneStepSDXLPipeline`, dift_sd.py:191-222): the reference's UNet forward never evaluates the "text_time" added-
    condition embedding, so only the SDXL topology (1 / 2 / 10 transformer layers, no attention at full resolution) and the
    two-encoder prompt (hidden_states[-2] of CLIP-L and OpenCLIP-bigG concatenated to 2048) are needed - both built.
"""
import glob
import json
import os

import torch

from law_of_vision_representation_in_mllms_amd.sd_engine import SdEngine
from law_of_vision_representation_in_mllms_amd.sd_weights import SD_SPECS, SchedulerSpec, SdSpec, TextSpec, UNetSpec, VaeSpec, synthetic_text, synthetic_unet, synthetic_vae
from law_of_vision_representation_in_mllms_amd.text_engine import ClipTextEngine
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder._vit_tower import _find_local_checkpoint


def _load_dir(path):
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        from safetensors.torch import load_file
        out = {}
        for f in st:
            if "fp16" in os.path.basename(f) and len(st) > 1:
                continue
            out.update(load_file(f))
        return out
    out = {}
    for f in sorted(glob.glob(os.path.join(path, "*.bin"))):
        out.update(torch.load(f, map_location="cpu"))
    if not out:
        raise OSError(f"no weights under {path}")
    return out


def _json(path):
    with open(path) as f:
        return json.load(f)


def _tuple(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


def spec_from_checkpoint(name, root, need_text=True, text_dir="text_encoder") -> tuple:
    u = _json(os.path.join(root, "unet", "config.json"))
    v = _json(os.path.join(root, "vae", "config.json"))
    s = _json(os.path.join(root, "scheduler", "scheduler_config.json"))
    t = _json(os.path.join(root, text_dir, "config.json")) if need_text else None
    nb = len(u["block_out_channels"])
    if u.get("class_embed_type"):
        raise NotImplementedError("class-conditional UNets are not built on the MI355X path")
    tl = u.get("transformer_layers_per_block", 1)
    unet = UNetSpec(in_channels=u["in_channels"], block_out=tuple(u["block_out_channels"]), down_types=tuple(u["down_block_types"]),
                    up_types=tuple(u["up_block_types"]), layers_per_block=u["layers_per_block"], heads=_tuple(u["attention_head_dim"], nb),
                    cross_dim=u["cross_attention_dim"], groups=u["norm_num_groups"], eps=u.get("norm_eps", 1e-5),
                    linear_projection=bool(u.get("use_linear_projection", False)),
                    tlayers=tuple(tl) if isinstance(tl, (list, tuple)) else ((tl,) * nb if tl != 1 else ()))
    vae = VaeSpec(in_channels=v["in_channels"], block_out=tuple(v["block_out_channels"]), layers_per_block=v["layers_per_block"],
                  latent_channels=v["latent_channels"], groups=v["norm_num_groups"], scaling_factor=v.get("scaling_factor", 0.18215))
    sched = SchedulerSpec(num_train_timesteps=s["num_train_timesteps"], beta_start=s["beta_start"], beta_end=s["beta_end"],
                          beta_schedule=s["beta_schedule"])
    if t is None:
        return SdSpec(name, unet, vae, sched, text_len=1), None
    text = TextSpec(vocab=t["vocab_size"], d=t["hidden_size"], mlp=t["intermediate_size"], layers=t["num_hidden_layers"],
                    heads=t["num_attention_heads"], max_pos=t["max_position_embeddings"], act=t.get("hidden_act", "quick_gelu"),
                    eps=t.get("layer_norm_eps", 1e-5))
    return SdSpec(name, unet, vae, sched, text_len=text.max_pos), text


_SYNTH_TEXT = {"runwayml/stable-diffusion-v1-5": TextSpec(),
               "stabilityai/stable-diffusion-2-1": TextSpec(d=1024, mlp=4096, layers=23, heads=16, act="gelu"),
               "stabilityai/stable-diffusion-xl-base-1.0": TextSpec()}
_SYNTH_TEXT_2 = {"stabilityai/stable-diffusion-xl-base-1.0": TextSpec(d=1280, mlp=5120, layers=32, heads=20, act="gelu")}


class SDFeaturizer:
    def __init__(self, sd_id='stabilityai/stable-diffusion-2-1', device=None, synthetic=None):
        self.sd_id = sd_id
        self.is_xl = 'xl' in sd_id                                            # the reference's own switch (dift_sd.py:227)
        self.device = torch.device(device if device is not None else "cuda")
        synthetic = os.environ.get("VISREP_SYNTHETIC_WEIGHTS") == "1" if synthetic is None else synthetic
        root = None if synthetic else _find_local_checkpoint(sd_id)
        self.tokenizer = self.tokenizer_2 = None
        self.text_2 = None
        if root is not None:
            self.spec, self.text_spec = spec_from_checkpoint(sd_id, root)
            self._wu, self._wv = _load_dir(os.path.join(root, "unet")), _load_dir(os.path.join(root, "vae"))
            wt = {k.replace("text_model.", "", 1): v for k, v in _load_dir(os.path.join(root, "text_encoder")).items()}
            from transformers import CLIPTokenizer
            self.tokenizer = CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer"))
            if self.is_xl:
                _, ts2 = spec_from_checkpoint(sd_id, root, text_dir="text_encoder_2")
                wt2 = {k.replace("text_model.", "", 1): v for k, v in _load_dir(os.path.join(root, "text_encoder_2")).items()}
                self.text_2 = ClipTextEngine(ts2, wt2, self.device)
                self.tokenizer_2 = CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer_2"))
        elif synthetic:
            if sd_id not in SD_SPECS:
                raise KeyError(f"no built-in architecture for {sd_id}")
            self.spec, self.text_spec = SD_SPECS[sd_id], _SYNTH_TEXT[sd_id]
            n_up = len(self.spec.unet.block_out)
            self._wu, self._wv = synthetic_unet(self.spec.unet, 21, n_up_blocks=n_up), synthetic_vae(self.spec.vae, 22)
            wt = synthetic_text(self.text_spec, 23)
            if self.is_xl:
                self.text_2 = ClipTextEngine(_SYNTH_TEXT_2[sd_id], synthetic_text(_SYNTH_TEXT_2[sd_id], 24), self.device)
        else:
            raise OSError(f"{sd_id} is not a local diffusers checkpoint directory and is not in the offline HF cache "
                          "(set VISREP_SYNTHETIC_WEIGHTS=1 for deterministic random-init weights)")
        self.text = ClipTextEngine(self.text_spec, wt, self.device)
        self._engines = {}
        self._prompt_cache = {}
        self.dtype = torch.bfloat16

    def _engine(self, up_ft_index) -> SdEngine:
        if up_ft_index not in self._engines:
            self._engines[up_ft_index] = SdEngine(self.spec, self._wu, self._wv, self.device, up_ft_index=up_ft_index)
        return self._engines[up_ft_index]

    def tokenize(self, prompt: str, second: bool = False) -> torch.Tensor:
        L = self.text_spec.max_pos
        tok = self.tokenizer_2 if second else self.tokenizer
        if tok is not None:                                 # pipe.encode_prompt: padding="max_length", truncation
            return tok(prompt, padding="max_length", max_length=L, truncation=True, return_tensors="pt").input_ids
        v = self.text_spec.vocab                            # synthetic weights: byte-level stand-in, <bos> bytes <eos>-padding
        body = [b % (v - 2) for b in prompt.encode("utf-8")][: L - 2]
        return torch.tensor([[v - 2] + body + [v - 1] * (L - 1 - len(body))], dtype=torch.long)

    def encode_prompt(self, prompt: str) -> torch.Tensor:
        """[1, L, cross_dim] bf16 - what pipe.encode_prompt(...) returns for one prompt (dift_sd.py:258-263)."""
        if prompt not in self._prompt_cache:
            if self.is_xl:   # pipeline_stable_diffusion_xl.py encode_prompt: hidden_states[-2] of both encoders, concatenated
                e1 = self.text.forward(self.tokenize(prompt), hidden_state=-2)
                e2 = self.text_2.forward(self.tokenize(prompt, second=True), hidden_state=-2)
                self._prompt_cache[prompt] = torch.cat([e1, e2], dim=-1).contiguous()
            else:
                self._prompt_cache[prompt] = self.text.forward(self.tokenize(prompt))
        return self._prompt_cache[prompt]

    @torch.no_grad()
    def forward(self, img_tensor, prompt, t=1, up_ft_index=0, ensemble_size=1, post_noise=None, ddim_noise=None):
        '''
        Args / Return as the reference: img_tensor [B, C, H, W]; returns [B, 1, c, h, w].squeeze() (dift_sd.py:240-276).
        '''
        eng = self._engine(up_ft_index)
        B = img_tensor.shape[0]
        tokens = eng.forward(img_tensor, self.encode_prompt(prompt), t=t, ensemble_size=ensemble_size, post_noise=post_noise,
                             ddim_noise=ddim_noise)                                                   # [B, h*w, c]
        f = 2 ** (len(self.spec.vae.block_out) - 1)
        lh, lw = img_tensor.shape[2] // f, img_tensor.shape[3] // f
        h = int(round((tokens.shape[1] * lh / lw) ** 0.5))
        w = tokens.shape[1] // h
        unet_ft = tokens.view(B, 1, h, w, tokens.shape[2]).permute(0, 1, 4, 2, 3)                     # a view: no NCHW copy is made
        return unet_ft.squeeze()
