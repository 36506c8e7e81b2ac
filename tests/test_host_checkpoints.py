"""The on-disk checkpoint route, executed (VERDICT r3 "what's weak" 1): directory discovery -> AutoConfig -> ViTSpec (incl. the
processor's crop size) -> safetensors / multi-shard .bin -> text-tower key filtering -> packed weights, for the three ViT families the
reference loads with `from_pretrained` (llava/model/multimodal_encoder/clip_encoder.py:22-27, dinov2_encoder.py:22-27,
siglip_encoder.py:22-27), and the diffusers-layout directory the diffusion featurizers read (unet/ vae/ scheduler/ text_encoder/).

Checkpoints are tiny random-init HF models written with `save_pretrained` into tmp_path; the towers are built with delay_load=True so
nothing here touches a GPU (tests/test_gpu_dropin.py holds the twin that runs a tower built from such a directory)."""
import json
import os
from types import SimpleNamespace

import pytest
import torch

from law_of_vision_representation_in_mllms_amd import vit_weights as VW
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder import _vit_tower as VT

transformers = pytest.importorskip("transformers")


def _args(**kw):
    return SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature="patch", **kw)


def tiny_hf_model(kind: str):
    """(HF model, tower class, expected family) - head width 64 so that the same directory also serves the GPU twin."""
    torch.manual_seed(3)
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.dinov2_encoder import DinoV2VisionTower
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.siglip_encoder import SigLipVisionTower
    if kind == "clip_vision":
        cfg = transformers.CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=42,
                                            patch_size=14, hidden_act="quick_gelu")
        return transformers.CLIPVisionModel(cfg), CLIPVisionTower, "clip"
    if kind == "clip_full":                                     # what openai/clip-vit-* actually ship: text tower + projections beside the vision tower
        vc = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=42, patch_size=14, hidden_act="quick_gelu")
        tc = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=99, max_position_embeddings=16)
        cfg = transformers.CLIPConfig(text_config=tc, vision_config=vc, projection_dim=32)
        return transformers.CLIPModel(cfg), CLIPVisionTower, "clip"
    if kind == "dinov2":
        cfg = transformers.Dinov2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, mlp_ratio=2, image_size=70, patch_size=14)
        return transformers.Dinov2Model(cfg), DinoV2VisionTower, "dinov2"
    if kind == "siglip":
        cfg = transformers.SiglipVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=48,
                                              patch_size=16)
        return transformers.SiglipVisionModel(cfg), SigLipVisionTower, "siglip"
    raise KeyError(kind)


def write_checkpoint(model, path, fmt: str, crop=None):
    os.makedirs(path, exist_ok=True)
    if fmt == "safetensors":
        model.save_pretrained(path, safe_serialization=True)
        assert any(f.endswith(".safetensors") for f in os.listdir(path))
    else:                                                       # two-shard .bin in the legacy naming (pytorch_model-0000k-of-00002.bin + index)
        model.config.save_pretrained(path)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        keys = sorted(sd)
        half = len(keys) // 2
        names = ["pytorch_model-00001-of-00002.bin", "pytorch_model-00002-of-00002.bin"]
        torch.save({k: sd[k] for k in keys[:half]}, os.path.join(path, names[0]))
        torch.save({k: sd[k] for k in keys[half:]}, os.path.join(path, names[1]))
        with open(os.path.join(path, "pytorch_model.bin.index.json"), "w") as fh:
            json.dump({"metadata": {}, "weight_map": {k: names[0 if i < half else 1] for i, k in enumerate(keys)}}, fh)
    if crop is not None:
        with open(os.path.join(path, "preprocessor_config.json"), "w") as fh:
            json.dump({"crop_size": crop, "do_center_crop": True, "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5]}, fh)


def assert_packed_equal(a, b):
    fa, fb = VW.flatten(a), VW.flatten(b)
    assert sorted(fa) == sorted(fb)
    for k in fa:
        assert torch.equal(fa[k], fb[k]), k


@pytest.mark.parametrize("fmt", ["safetensors", "bin2"])
@pytest.mark.parametrize("kind", ["clip_vision", "clip_full", "dinov2", "siglip"])
def test_tower_reads_a_checkpoint_directory(tmp_path, kind, fmt):
    model, Tower, family = tiny_hf_model(kind)
    path = str(tmp_path / kind)
    crop = {"dinov2": {"height": 42, "width": 42}}.get(kind)          # the DINOv2 processor's crop_size is a dict in HF's files
    write_checkpoint(model, path, fmt, crop)
    tower = Tower(path, _args(), delay_load=True)
    assert not tower.is_loaded and tower.cfg_only is not None          # delay_load: the config comes from the directory, nothing is built
    spec, w = tower._spec_and_weights()
    vc = model.config.vision_config if kind == "clip_full" else model.config
    # spec == config
    assert spec.family == family and (spec.d, spec.layers, spec.heads, spec.patch) == (vc.hidden_size, vc.num_hidden_layers, vc.num_attention_heads, vc.patch_size)
    assert spec.mlp == (int(vc.hidden_size * vc.mlp_ratio) if kind == "dinov2" else vc.intermediate_size)
    if kind == "dinov2":
        assert spec.image_size == 42 and spec.pos_grid == 5               # built at the PROCESSOR's crop size; the native 5 x 5 grid is kept aside
        assert w["pos"].shape[0] == 1 + 9 and w["pos_native"].shape[0] == 1 + 25
    else:
        assert spec.image_size == vc.image_size
    # weights == pack_hf_state_dict(model.state_dict()) with the text tower / projections of a full CLIP checkpoint filtered out
    sd = {k: v for k, v in model.state_dict().items() if not k.startswith(("text_model.", "logit_", "text_projection", "visual_projection"))}
    assert_packed_equal(w, VW.pack_hf_state_dict(sd, spec))
    if kind == "clip_full":
        assert any(k.startswith("text_model.") for k in model.state_dict())     # the filter had something to do


def test_processor_crop_forms_and_missing_directory(tmp_path):
    for body, want in (({"crop_size": 224}, 224), ({"crop_size": {"height": 336, "width": 336}}, 336), ({"crop_size": {"shortest_edge": 518}}, 518),
                       ({"size": 256}, None)):
        p = tmp_path / f"c{want}"
        p.mkdir(exist_ok=True)
        (p / "preprocessor_config.json").write_text(json.dumps(body))
        assert VT._processor_crop(str(p)) == want
    assert VT._processor_crop(str(tmp_path)) is None                     # no file
    (tmp_path / "preprocessor_config.json").write_text("{not json")
    assert VT._processor_crop(str(tmp_path)) is None
    assert VT._find_local_checkpoint(str(tmp_path / "nowhere")) is None
    (tmp_path / "empty").mkdir()
    assert VT._find_local_checkpoint(str(tmp_path / "empty")) is None    # a directory without config.json is not a checkpoint
    with pytest.raises(OSError, match="no weights"):
        VT._load_state_dict(str(tmp_path / "empty"))
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    with pytest.raises(OSError, match="not a local checkpoint"):            # delay_load reads the config first: same error as from_pretrained
        CLIPVisionTower(str(tmp_path / "nowhere"), _args(), delay_load=True)
    known = CLIPVisionTower("openai/clip-vit-large-patch14", _args(), delay_load=True)     # a known name offline: built-in architecture for the config ...
    assert known.hidden_size == 1024 and known.num_patches == 256
    with pytest.raises(OSError, match="VISREP_SYNTHETIC_WEIGHTS"):          # ... but no weights without the explicit synthetic switch
        known._spec_and_weights()


# ------------------------------------------------------------------------------------------------ diffusers-layout directory
def write_diffusers_dir(root, spec, text_spec, wu, wv, wt, fmt="safetensors"):
    """unet/ vae/ scheduler/ text_encoder/ with the config.json fields diffusers writes and the weights under diffusers' file names."""
    from safetensors.torch import save_file
    u, v, s = spec.unet, spec.vae, spec.sched
    cfgs = {
        "unet": {"_class_name": "UNet2DConditionModel", "in_channels": u.in_channels, "block_out_channels": list(u.block_out),
                 "down_block_types": list(u.down_types), "up_block_types": list(u.up_types), "layers_per_block": u.layers_per_block,
                 "attention_head_dim": list(u.heads), "cross_attention_dim": u.cross_dim, "norm_num_groups": u.groups, "norm_eps": u.eps,
                 "use_linear_projection": u.linear_projection, "class_embed_type": None},
        "vae": {"_class_name": "AutoencoderKL", "in_channels": v.in_channels, "block_out_channels": list(v.block_out), "layers_per_block": v.layers_per_block,
                "latent_channels": v.latent_channels, "norm_num_groups": v.groups, "scaling_factor": v.scaling_factor},
        "text_encoder": {"model_type": "clip_text_model", "vocab_size": text_spec.vocab, "hidden_size": text_spec.d, "intermediate_size": text_spec.mlp,
                         "num_hidden_layers": text_spec.layers, "num_attention_heads": text_spec.heads, "max_position_embeddings": text_spec.max_pos,
                         "hidden_act": text_spec.act, "layer_norm_eps": text_spec.eps},
    }
    weights = {"unet": wu, "vae": wv, "text_encoder": {"text_model." + k: t for k, t in wt.items()}}
    for sub in ("unet", "vae", "text_encoder"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        with open(os.path.join(root, sub, "config.json"), "w") as fh:
            json.dump(cfgs[sub], fh)
        sd = {k: t.contiguous() for k, t in weights[sub].items()}
        base = "model" if sub == "text_encoder" else "diffusion_pytorch_model"
        if fmt == "safetensors":
            save_file(sd, os.path.join(root, sub, base + ".safetensors"))
            if sub == "unet":                                    # the half-precision twin diffusers repos carry: must be skipped, not merged
                save_file({k: t.half() * 0 for k, t in sd.items()}, os.path.join(root, sub, base + ".fp16.safetensors"))
        else:
            torch.save(sd, os.path.join(root, sub, base + ".bin"))
    os.makedirs(os.path.join(root, "scheduler"), exist_ok=True)
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as fh:
        json.dump({"_class_name": "DDIMScheduler", "num_train_timesteps": s.num_train_timesteps, "beta_start": s.beta_start, "beta_end": s.beta_end,
                   "beta_schedule": s.beta_schedule}, fh)
    with open(os.path.join(root, "model_index.json"), "w") as fh:
        json.dump({"_class_name": "StableDiffusionPipeline"}, fh)


def tiny_sd_checkpoint():
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    spec = SW.tiny_sd_spec()
    ts = SW.TextSpec(vocab=99, d=spec.unet.cross_dim, mlp=128, layers=2, heads=1, max_pos=11, act="quick_gelu")     # d = the UNet's cross-attention width
    return spec, ts, SW.synthetic_unet(spec.unet, 21, n_up_blocks=len(spec.unet.block_out)), SW.synthetic_vae(spec.vae, 22), SW.synthetic_text(ts, 23)


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_diffusers_directory_gives_the_spec_and_the_weights(tmp_path, fmt):
    """dift_sd.spec_from_checkpoint / _load_dir on a diffusers-layout directory (what SDFeaturizer does with a local SD checkpoint,
    reference: diffLVLM/src/models/dift_sd.py:226-243 `from_pretrained(sd_id, subfolder=...)`)."""
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM.src.models import dift_sd as DS
    spec, ts, wu, wv, wt = tiny_sd_checkpoint()
    root = str(tmp_path / "tiny-sd")
    write_diffusers_dir(root, spec, ts, wu, wv, wt, fmt)
    assert DS._find_local_checkpoint(root) == root                     # a diffusers directory has model_index.json, not config.json (found by this test: r4)
    got, got_text = DS.spec_from_checkpoint(spec.name, root)
    assert got.unet == spec.unet and got.vae == spec.vae and got.sched == spec.sched
    assert got_text == ts and got.text_len == ts.max_pos
    for sub, want in (("unet", wu), ("vae", wv)):
        sd = DS._load_dir(os.path.join(root, sub))
        assert sorted(sd) == sorted(want)
        for k in want:
            assert torch.equal(sd[k], want[k]), (sub, k)            # the .fp16 twin (all zeros here) was not merged over the real file
    te = {k.replace("text_model.", "", 1): v for k, v in DS._load_dir(os.path.join(root, "text_encoder")).items()}
    assert sorted(te) == sorted(wt) and all(torch.equal(te[k], wt[k]) for k in wt)
    nt, none = DS.spec_from_checkpoint(spec.name, root, need_text=False)
    assert none is None and nt.unet == spec.unet
    with pytest.raises(OSError, match="no weights"):
        os.makedirs(os.path.join(root, "nothing"), exist_ok=True)
        DS._load_dir(os.path.join(root, "nothing"))
