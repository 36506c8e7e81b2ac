"""Architecture specs and parameter tables of the Stable-Diffusion feature towers.

The reference loads `UNet2DConditionModel` / `AutoencoderKL` checkpoints through diffusers (`dift_sd.py:224-236`); the
parameter names below are the diffusers state-dict names, so a local checkpoint's `unet/` and `vae/` safetensors load
without renaming.  Only what the feature path executes is listed: the VAE *encoder* (+ quant_conv) and the UNet up to the
up-block whose output is captured (`dift_sd.py:118-150` stops after `max(up_ft_indices)`).

No network here, so `synthetic_*` produce deterministic random-init parameters of a given architecture
(numpy RandomState: version-stable, regenerated identically on the GPU box).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class UNetSpec:
    """UNet2DConditionModel config subset (SD1.5: stable-diffusion-v1-5/unet/config.json)."""
    in_channels: int = 4
    block_out: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    layers_per_block: int = 2
    heads: Tuple[int, ...] = (8, 8, 8, 8)          # diffusers' `attention_head_dim` is the HEAD COUNT for SD1.x
    cross_dim: int = 768
    groups: int = 32
    eps: float = 1e-5
    linear_projection: bool = False                # SD2.1 / SDXL: proj_in / proj_out are Linear
    tlayers: Tuple[int, ...] = ()                  # transformer_layers_per_block (SDXL: 1, 2, 10); () = one everywhere

    @property
    def temb_dim(self):
        return 4 * self.block_out[0]

    def depth(self, block: int) -> int:
        """BasicTransformerBlocks per Transformer2DModel of down block `block` (up block i uses depth(n-1-i), mid depth(n-1))."""
        return self.tlayers[block] if self.tlayers else 1


@dataclass(frozen=True)
class VaeSpec:
    """AutoencoderKL encoder half (SD1.5 vae/config.json)."""
    in_channels: int = 3
    block_out: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    groups: int = 32
    scaling_factor: float = 0.18215
    quant_conv: bool = True                        # SD3's 16-channel VAE: use_quant_conv = False


@dataclass(frozen=True)
class SchedulerSpec:
    """DDIMScheduler fields `add_noise` depends on (scheduling_ddim.py:471-495)."""
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "scaled_linear"

    def alphas_cumprod(self) -> torch.Tensor:
        if self.beta_schedule == "scaled_linear":
            betas = torch.linspace(self.beta_start ** 0.5, self.beta_end ** 0.5, self.num_train_timesteps, dtype=torch.float32) ** 2
        elif self.beta_schedule == "linear":
            betas = torch.linspace(self.beta_start, self.beta_end, self.num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(f"{self.beta_schedule} is not implemented")
        return torch.cumprod(1.0 - betas, dim=0)


@dataclass(frozen=True)
class SdSpec:
    name: str
    unet: UNetSpec = field(default_factory=UNetSpec)
    vae: VaeSpec = field(default_factory=VaeSpec)
    sched: SchedulerSpec = field(default_factory=SchedulerSpec)
    text_len: int = 77


SD_SPECS: Dict[str, SdSpec] = {
    "runwayml/stable-diffusion-v1-5": SdSpec("runwayml/stable-diffusion-v1-5"),
    "stabilityai/stable-diffusion-2-1": SdSpec(
        "stabilityai/stable-diffusion-2-1",
        unet=UNetSpec(heads=(5, 10, 20, 20), cross_dim=1024, linear_projection=True)),
    # SDXL base: 3 blocks, no attention at full resolution, 1 / 2 / 10 transformer layers, two text encoders (768 + 1280).
    # The reference's UNet forward never calls `add_embedding` (dift_sd.py:70-88 has no aug-emb step), so the "text_time"
    # added-condition branch of the checkpoint is dead weight on this path.
    "stabilityai/stable-diffusion-xl-base-1.0": SdSpec(
        "stabilityai/stable-diffusion-xl-base-1.0",
        unet=UNetSpec(block_out=(320, 640, 1280), down_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                      up_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), heads=(5, 10, 20), cross_dim=2048,
                      linear_projection=True, tlayers=(1, 2, 10)),
        vae=VaeSpec(scaling_factor=0.13025)),
}


def tiny_sdxl_spec() -> SdSpec:
    """SDXL topology in miniature: first block without attention, deeper transformers lower down, Linear projections."""
    return SdSpec("tiny-sdxl",
                  unet=UNetSpec(block_out=(64, 128, 128), down_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                                up_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), heads=(1, 2, 2), cross_dim=64,
                                linear_projection=True, tlayers=(1, 2, 3)),
                  vae=VaeSpec(block_out=(64, 64, 128), layers_per_block=1, scaling_factor=0.13025), text_len=11)


def tiny_sd_spec(name="tiny-sd", linear_projection=False) -> SdSpec:
    """Same topology as SD1.5, every width a multiple of 64 (the GEMM kernels' granularity), head widths 32 / 64."""
    return SdSpec(name,
                  unet=UNetSpec(block_out=(64, 128, 128), down_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                                up_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"), heads=(2, 2, 2), cross_dim=64,
                                linear_projection=linear_projection),
                  vae=VaeSpec(block_out=(64, 64, 128), layers_per_block=1), text_len=11)


# ----------------------------------------------------------------------------------------------- parameter tables
def _resnet(p, cin, cout, temb):
    t = [(f"{p}.norm1.weight", (cin,)), (f"{p}.norm1.bias", (cin,)),
         (f"{p}.conv1.weight", (cout, cin, 3, 3)), (f"{p}.conv1.bias", (cout,))]
    if temb:
        t += [(f"{p}.time_emb_proj.weight", (cout, temb)), (f"{p}.time_emb_proj.bias", (cout,))]
    t += [(f"{p}.norm2.weight", (cout,)), (f"{p}.norm2.bias", (cout,)),
          (f"{p}.conv2.weight", (cout, cout, 3, 3)), (f"{p}.conv2.bias", (cout,))]
    if cin != cout:
        t += [(f"{p}.conv_shortcut.weight", (cout, cin, 1, 1)), (f"{p}.conv_shortcut.bias", (cout,))]
    return t


def _transformer(p, d, cross, linear, depth=1):
    proj = (d, d) if linear else (d, d, 1, 1)
    t = [(f"{p}.norm.weight", (d,)), (f"{p}.norm.bias", (d,)), (f"{p}.proj_in.weight", proj), (f"{p}.proj_in.bias", (d,))]
    for k in range(depth):
        t += _basic_block(f"{p}.transformer_blocks.{k}", d, cross)
    return t + [(f"{p}.proj_out.weight", proj), (f"{p}.proj_out.bias", (d,))]


def _basic_block(b, d, cross):
    return [(f"{b}.norm1.weight", (d,)), (f"{b}.norm1.bias", (d,)),
            (f"{b}.attn1.to_q.weight", (d, d)), (f"{b}.attn1.to_k.weight", (d, d)), (f"{b}.attn1.to_v.weight", (d, d)),
            (f"{b}.attn1.to_out.0.weight", (d, d)), (f"{b}.attn1.to_out.0.bias", (d,)),
            (f"{b}.norm2.weight", (d,)), (f"{b}.norm2.bias", (d,)),
            (f"{b}.attn2.to_q.weight", (d, d)), (f"{b}.attn2.to_k.weight", (d, cross)), (f"{b}.attn2.to_v.weight", (d, cross)),
            (f"{b}.attn2.to_out.0.weight", (d, d)), (f"{b}.attn2.to_out.0.bias", (d,)),
            (f"{b}.norm3.weight", (d,)), (f"{b}.norm3.bias", (d,)),
            (f"{b}.ff.net.0.proj.weight", (8 * d, d)), (f"{b}.ff.net.0.proj.bias", (8 * d,)),
            (f"{b}.ff.net.2.weight", (d, 4 * d)), (f"{b}.ff.net.2.bias", (d,))]


def up_block_plan(u: UNetSpec, i: int):
    """(resnet input widths incl. skip, output width, has_attention, has_upsampler) of up block i (unet_2d_condition.py)."""
    rev = tuple(reversed(u.block_out))
    n = len(rev)
    out, prev, inp = rev[i], rev[max(i - 1, 0)], rev[min(i + 1, n - 1)]
    L = u.layers_per_block + 1
    cins = [(prev if j == 0 else out) + (inp if j == L - 1 else out) for j in range(L)]
    return cins, out, u.up_types[i].startswith("CrossAttn"), i != n - 1


def unet_param_table(u: UNetSpec, n_up_blocks: int = 1) -> List[Tuple[str, tuple]]:
    c0, T = u.block_out[0], u.temb_dim
    t = [("conv_in.weight", (c0, u.in_channels, 3, 3)), ("conv_in.bias", (c0,)),
         ("time_embedding.linear_1.weight", (T, c0)), ("time_embedding.linear_1.bias", (T,)),
         ("time_embedding.linear_2.weight", (T, T)), ("time_embedding.linear_2.bias", (T,))]
    cin = c0
    for i, cout in enumerate(u.block_out):
        for j in range(u.layers_per_block):
            t += _resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, T)
            if u.down_types[i].startswith("CrossAttn"):
                t += _transformer(f"down_blocks.{i}.attentions.{j}", cout, u.cross_dim, u.linear_projection, u.depth(i))
        if i != len(u.block_out) - 1:
            t += [(f"down_blocks.{i}.downsamplers.0.conv.weight", (cout, cout, 3, 3)), (f"down_blocks.{i}.downsamplers.0.conv.bias", (cout,))]
        cin = cout
    cm = u.block_out[-1]
    nb = len(u.block_out)
    t += _resnet("mid_block.resnets.0", cm, cm, T) + _transformer("mid_block.attentions.0", cm, u.cross_dim, u.linear_projection, u.depth(nb - 1))
    t += _resnet("mid_block.resnets.1", cm, cm, T)
    for i in range(n_up_blocks):
        cins, out, attn, ups = up_block_plan(u, i)
        for j, ci in enumerate(cins):
            t += _resnet(f"up_blocks.{i}.resnets.{j}", ci, out, T)
            if attn:
                t += _transformer(f"up_blocks.{i}.attentions.{j}", out, u.cross_dim, u.linear_projection, u.depth(nb - 1 - i))
        if ups:
            t += [(f"up_blocks.{i}.upsamplers.0.conv.weight", (out, out, 3, 3)), (f"up_blocks.{i}.upsamplers.0.conv.bias", (out,))]
    return t


def vae_param_table(v: VaeSpec) -> List[Tuple[str, tuple]]:
    c0 = v.block_out[0]
    t = [("encoder.conv_in.weight", (c0, v.in_channels, 3, 3)), ("encoder.conv_in.bias", (c0,))]
    cin = c0
    for i, cout in enumerate(v.block_out):
        for j in range(v.layers_per_block):
            t += _resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, 0)
        if i != len(v.block_out) - 1:
            t += [(f"encoder.down_blocks.{i}.downsamplers.0.conv.weight", (cout, cout, 3, 3)),
                  (f"encoder.down_blocks.{i}.downsamplers.0.conv.bias", (cout,))]
        cin = cout
    cm = v.block_out[-1]
    t += _resnet("encoder.mid_block.resnets.0", cm, cm, 0)
    a = "encoder.mid_block.attentions.0"
    t += [(f"{a}.group_norm.weight", (cm,)), (f"{a}.group_norm.bias", (cm,))]
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        t += [(f"{a}.{n}.weight", (cm, cm)), (f"{a}.{n}.bias", (cm,))]
    t += _resnet("encoder.mid_block.resnets.1", cm, cm, 0)
    z = 2 * v.latent_channels
    t += [("encoder.conv_norm_out.weight", (cm,)), ("encoder.conv_norm_out.bias", (cm,)),
          ("encoder.conv_out.weight", (z, cm, 3, 3)), ("encoder.conv_out.bias", (z,))]
    if v.quant_conv:
        t += [("quant_conv.weight", (z, z, 1, 1)), ("quant_conv.bias", (z,))]
    return t


def _synthetic_fast(table, seed):
    """Same distributions drawn with a torch generator (10-20x faster than numpy's RandomState for the 1-3 G parameters of the SD /
    SDXL / SD3 architectures).  Used by throughput runs (VISREP_FAST_SYNTHETIC=1, the sweep) where the values only need to be
    deterministic within one torch version; fixtures and parity tests keep the version-stable numpy stream below."""
    import os
    dev = "cuda" if os.environ.get("VISREP_FAST_SYNTHETIC") == "cuda" and torch.cuda.is_available() else "cpu"   # "cuda": drawn on the GPU (the host RNG is the slow part), copied back
    g = torch.Generator(device=dev).manual_seed(seed)
    out = {}
    for name, shape in table:
        if name.endswith("bias"):
            out[name] = torch.randn(shape, generator=g, device=dev) * 0.05
        elif "norm" in name.split(".")[-2]:
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g, device=dev)
        else:
            out[name] = torch.randn(shape, generator=g, device=dev) * (1.0 / np.sqrt(int(np.prod(shape[1:]))))
    # the engines do their host-side folds (time embeddings, weight repacking) on CPU tensors: hand the values back to the host
    return {k: v.cpu() for k, v in out.items()} if dev == "cuda" else out


def _synthetic(table, seed):
    import os
    if os.environ.get("VISREP_FAST_SYNTHETIC") in ("1", "cuda"):
        return _synthetic_fast(table, seed)
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in table:
        if name.endswith("bias"):
            w = rs.standard_normal(shape) * 0.05
        elif "norm" in name.split(".")[-2]:          # affine of a normalisation layer (not e.g. `norm1.linear.weight`)
            w = 1.0 + 0.1 * rs.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            w = rs.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
        out[name] = torch.from_numpy(w.astype(np.float32))
    return out


def synthetic_unet(u: UNetSpec, seed: int, n_up_blocks: int = 1):
    return _synthetic(unet_param_table(u, n_up_blocks), seed)


def synthetic_vae(v: VaeSpec, seed: int):
    w = _synthetic(vae_param_table(v), seed)
    # keep the posterior log-variance moderate so exp(0.5 * logvar) * noise stays O(1) like a trained VAE's
    z = v.latent_channels
    w["quant_conv.bias" if v.quant_conv else "encoder.conv_out.bias"][z:] -= 2.0
    return w


# ----------------------------------------------------------------------------------------------- CLIP text encoder
@dataclass(frozen=True)
class TextSpec:
    """HF CLIPTextConfig subset (SD1.5 text_encoder/config.json = openai/clip-vit-large-patch14 text tower)."""
    vocab: int = 49408
    d: int = 768
    mlp: int = 3072
    layers: int = 12
    heads: int = 12
    max_pos: int = 77
    act: str = "quick_gelu"
    eps: float = 1e-5


def text_param_table(s: TextSpec) -> List[Tuple[str, tuple]]:
    t = [("embeddings.token_embedding.weight", (s.vocab, s.d)), ("embeddings.position_embedding.weight", (s.max_pos, s.d))]
    for i in range(s.layers):
        p = f"encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            t += [(f"{p}.self_attn.{n}.weight", (s.d, s.d)), (f"{p}.self_attn.{n}.bias", (s.d,))]
        t += [(f"{p}.layer_norm1.weight", (s.d,)), (f"{p}.layer_norm1.bias", (s.d,)),
              (f"{p}.mlp.fc1.weight", (s.mlp, s.d)), (f"{p}.mlp.fc1.bias", (s.mlp,)),
              (f"{p}.mlp.fc2.weight", (s.d, s.mlp)), (f"{p}.mlp.fc2.bias", (s.d,)),
              (f"{p}.layer_norm2.weight", (s.d,)), (f"{p}.layer_norm2.bias", (s.d,))]
    return t + [("final_layer_norm.weight", (s.d,)), ("final_layer_norm.bias", (s.d,))]


def synthetic_text(s: TextSpec, seed: int):
    w = _synthetic(text_param_table(s), seed)
    w["embeddings.token_embedding.weight"] *= s.d ** 0.5 * 0.5          # O(1) embeddings like a trained table after LN
    w["embeddings.position_embedding.weight"] *= s.d ** 0.5 * 0.2
    return w


def tiny_text_spec(act="quick_gelu", layers=2, max_pos=16) -> TextSpec:
    return TextSpec(vocab=99, d=128, mlp=256, layers=layers, heads=2, max_pos=max_pos, act=act)


# ----------------------------------------------------------------------------------------------- DiT (facebook/DiT-XL-2-512)
@dataclass(frozen=True)
class DiTCoreSpec:
    """DiTTransformer2DModel config subset (dit_transformer_2d.py:72-90; DiT-XL/2: 28 layers, 16 heads x 72)."""
    heads: int = 16
    head_dim: int = 72
    in_channels: int = 4
    layers: int = 28
    sample_size: int = 64          # latent side the sincos table is built for (512-px checkpoint)
    patch: int = 2
    eps: float = 1e-5
    num_classes: int = 1000

    @property
    def d(self):
        return self.heads * self.head_dim


@dataclass(frozen=True)
class DiTSpec:
    name: str
    core: DiTCoreSpec = field(default_factory=DiTCoreSpec)
    vae: VaeSpec = field(default_factory=VaeSpec)
    # DiT-XL-2-512/scheduler/scheduler_config.json: linear betas 1e-4 .. 0.02
    sched: SchedulerSpec = field(default_factory=lambda: SchedulerSpec(beta_start=0.0001, beta_end=0.02, beta_schedule="linear"))


DIT_SPECS: Dict[str, DiTSpec] = {"facebook/DiT-XL-2-512": DiTSpec("facebook/DiT-XL-2-512")}


def tiny_dit_spec() -> DiTSpec:
    """The real head width (72, zero-padded to 128 on the device) at a model width that is a multiple of 64."""
    return DiTSpec("tiny-dit", core=DiTCoreSpec(heads=8, head_dim=72, layers=2, sample_size=8, num_classes=10),
                   vae=VaeSpec(block_out=(64, 64, 128), layers_per_block=1))


def dit_param_table(c: DiTCoreSpec, n_layers: int = None) -> List[Tuple[str, tuple]]:
    D = c.d
    t = [("pos_embed.proj.weight", (D, c.in_channels, c.patch, c.patch)), ("pos_embed.proj.bias", (D,))]
    for i in range(c.layers if n_layers is None else n_layers):
        p = f"transformer_blocks.{i}"
        e = f"{p}.norm1.emb"
        t += [(f"{e}.timestep_embedder.linear_1.weight", (D, 256)), (f"{e}.timestep_embedder.linear_1.bias", (D,)),
              (f"{e}.timestep_embedder.linear_2.weight", (D, D)), (f"{e}.timestep_embedder.linear_2.bias", (D,)),
              (f"{e}.class_embedder.embedding_table.weight", (c.num_classes + 1, D)),       # dropped by the reference's override
              (f"{p}.norm1.linear.weight", (6 * D, D)), (f"{p}.norm1.linear.bias", (6 * D,))]
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            t += [(f"{p}.attn1.{n}.weight", (D, D)), (f"{p}.attn1.{n}.bias", (D,))]
        t += [(f"{p}.ff.net.0.proj.weight", (4 * D, D)), (f"{p}.ff.net.0.proj.bias", (4 * D,)),
              (f"{p}.ff.net.2.weight", (D, 4 * D)), (f"{p}.ff.net.2.bias", (D,))]
    return t


def synthetic_dit(c: DiTCoreSpec, seed: int, n_layers: int = None):
    w = _synthetic(dit_param_table(c, n_layers), seed)
    for k in w:                       # adaLN modulation: O(0.3) shifts / scales / gates instead of O(1) so depth stays tame
        if k.endswith("norm1.linear.weight"):
            w[k] *= 0.3
    return w


# ----------------------------------------------------------------------------------------------- SD3 (MMDiT)
@dataclass(frozen=True)
class Sd3CoreSpec:
    """SD3Transformer2DModel config subset (transformer_sd3.py:57-70; SD3-medium: 24 layers, 24 heads x 64 = 1536)."""
    heads: int = 24
    head_dim: int = 64
    in_channels: int = 16
    layers: int = 24
    sample_size: int = 128
    patch: int = 2
    joint_dim: int = 4096          # width of the prompt embeddings (CLIP-L|CLIP-G padded to 4096, T5 rows)
    pooled_dim: int = 2048         # pooled CLIP-L (768) | CLIP-G (1280) text embeds
    pos_max: int = 192             # pos_embed_max_size: side of the stored sincos table, centre-cropped per input

    @property
    def d(self):
        return self.heads * self.head_dim


@dataclass(frozen=True)
class Sd3Spec:
    name: str
    core: Sd3CoreSpec = field(default_factory=Sd3CoreSpec)
    vae: VaeSpec = field(default_factory=lambda: VaeSpec(latent_channels=16, scaling_factor=1.5305, quant_conv=False))
    sched: SchedulerSpec = field(default_factory=SchedulerSpec)        # unused: flow-matching add_noise needs no table


SD3_SPECS: Dict[str, Sd3Spec] = {"stabilityai/stable-diffusion-3-medium-diffusers": Sd3Spec("stabilityai/stable-diffusion-3-medium-diffusers")}


def tiny_sd3_spec() -> Sd3Spec:
    return Sd3Spec("tiny-sd3", core=Sd3CoreSpec(heads=2, head_dim=64, layers=3, sample_size=8, joint_dim=64, pooled_dim=64, pos_max=12),
                   vae=VaeSpec(block_out=(64, 64, 128), layers_per_block=1, latent_channels=16, scaling_factor=1.5305, quant_conv=False))


def sd3_param_table(c: Sd3CoreSpec, n_layers: int = None) -> List[Tuple[str, tuple]]:
    D = c.d
    t = [("pos_embed.proj.weight", (D, c.in_channels, c.patch, c.patch)), ("pos_embed.proj.bias", (D,)),
         ("time_text_embed.timestep_embedder.linear_1.weight", (D, 256)), ("time_text_embed.timestep_embedder.linear_1.bias", (D,)),
         ("time_text_embed.timestep_embedder.linear_2.weight", (D, D)), ("time_text_embed.timestep_embedder.linear_2.bias", (D,)),
         ("time_text_embed.text_embedder.linear_1.weight", (D, c.pooled_dim)), ("time_text_embed.text_embedder.linear_1.bias", (D,)),
         ("time_text_embed.text_embedder.linear_2.weight", (D, D)), ("time_text_embed.text_embedder.linear_2.bias", (D,)),
         ("context_embedder.weight", (D, c.joint_dim)), ("context_embedder.bias", (D,))]
    n = c.layers if n_layers is None else n_layers
    for i in range(n):
        p = f"transformer_blocks.{i}"
        last = i == c.layers - 1                                   # context_pre_only
        t += [(f"{p}.norm1.linear.weight", (6 * D, D)), (f"{p}.norm1.linear.bias", (6 * D,)),
              (f"{p}.norm1_context.linear.weight", ((2 if last else 6) * D, D)), (f"{p}.norm1_context.linear.bias", ((2 if last else 6) * D,))]
        names = ["to_q", "to_k", "to_v", "add_k_proj", "add_v_proj", "add_q_proj", "to_out.0"] + ([] if last else ["to_add_out"])
        for nme in names:
            t += [(f"{p}.attn.{nme}.weight", (D, D)), (f"{p}.attn.{nme}.bias", (D,))]
        for ff in ["ff"] + ([] if last else ["ff_context"]):
            t += [(f"{p}.{ff}.net.0.proj.weight", (4 * D, D)), (f"{p}.{ff}.net.0.proj.bias", (4 * D,)),
                  (f"{p}.{ff}.net.2.weight", (D, 4 * D)), (f"{p}.{ff}.net.2.bias", (D,))]
    return t


def synthetic_sd3(c: Sd3CoreSpec, seed: int, n_layers: int = None):
    w = _synthetic(sd3_param_table(c, n_layers), seed)
    for k in w:
        if k.endswith(("norm1.linear.weight", "norm1_context.linear.weight")):
            w[k] *= 0.3
    return w
