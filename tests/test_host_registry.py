"""CPU checks of the plug-in surface: registry keys and routing, error behaviour, config-only towers, weight packing."""
from types import SimpleNamespace

import pytest
import torch

from law_of_vision_representation_in_mllms_amd import vit_weights as VW
from law_of_vision_representation_in_mllms_amd.llava.model import llava_arch as LA
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder import builder as B
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_projector.builder import build_vision_projector

REFERENCE_REGISTRY_KEYS = [           # llava/model/llava_arch.py:29-40 of the reference
    'openai/clip-vit-large-patch14-336', 'google/siglip-base-patch16-224', 'laion/CLIP-ViT-L-14-laion2B-s32B-b82K',
    'stabilityai/stable-diffusion-2-1', 'runwayml/stable-diffusion-v1-5', 'lambdalabs/sd-image-variations-diffusers',
    'facebook/dinov2-large', 'stabilityai/stable-diffusion-xl-base-1.0', 'feature', 'facebook/DiT-XL-2-512',
    'stabilityai/stable-diffusion-3-medium-diffusers', 'openai/clip-vit-large-patch14']


def test_registry_has_the_reference_keys_and_routing():
    assert list(LA.build_function_mapping) == REFERENCE_REGISTRY_KEYS
    assert LA.build_function_mapping['openai/clip-vit-large-patch14-336'] is B.build_vision_tower
    assert LA.build_function_mapping['facebook/dinov2-large'] is B.build_dinov2_vision_tower
    assert LA.build_function_mapping['google/siglip-base-patch16-224'] is B.build_siglip_vision_tower
    assert LA.build_function_mapping['feature'](None) == 'feature'


def test_unknown_tower_raises_like_the_reference():
    with pytest.raises(ValueError, match="Unknown vision tower"):
        B.build_vision_tower(SimpleNamespace(mm_vision_tower="not/a-model", mm_vision_select_layer=-2))
    with pytest.raises(KeyError):
        LA.VisionEncoderStack(SimpleNamespace(mm_vision_tower="nope", mm_vision_select_layer=-2, mm_projector_type="linear", hidden_size=128))
    # diffusion towers: every featurizer of the reference's table is built; unknown ids fail like the reference (KeyError)
    dargs = dict(up_ft_index=0, t=100, prompt="", ensemble_size=1, img_size=768)
    with pytest.raises(KeyError):
        B.build_diffusion_vision_tower(SimpleNamespace(vision_tower='unknown/model', **dargs))


def test_diffusion_spec_tables_and_image_processor():
    from PIL import Image
    import numpy as np
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM import diffusion_encoder as DE
    assert DE.feature_hid_size_mapping['runwayml/stable-diffusion-v1-5'] == 1280
    assert sorted(DE.build_featurelizer_mapping) == sorted(['lambdalabs/sd-image-variations-diffusers', 'stabilityai/stable-diffusion-2-1',
                                                            'runwayml/stable-diffusion-v1-5', 'stabilityai/stable-diffusion-xl-base-1.0',
                                                            'facebook/DiT-XL-2-512', 'stabilityai/stable-diffusion-3-medium-diffusers'])
    assert DE.feature_hid_size_mapping['runwayml/stable-diffusion-v1-5_feature'] == 1280 and len(DE.feature_hid_size_mapping) == 7
    u = SW.SD_SPECS['runwayml/stable-diffusion-v1-5'].unet
    table = dict(SW.unet_param_table(u, n_up_blocks=1))
    assert table['up_blocks.0.resnets.2.conv1.weight'] == (1280, 2560, 3, 3)          # 1280 + skip 1280
    assert table['down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight'] == (320, 768)
    assert sum(int(np.prod(s)) for s in table.values()) == 510954240                   # UNet up to up_blocks[0] (full SD1.5 UNet: 859.5 M)
    assert SW.up_block_plan(u, 1) == ([1280 + 1280, 1280 + 1280, 1280 + 640], 1280, True, True)
    ac = SW.SchedulerSpec().alphas_cumprod()
    assert abs(float(ac[261]) - 0.6557) < 1e-3 and abs(float(ac[0]) - 0.99915) < 1e-4
    img = Image.fromarray(np.full((10, 14, 3), 255, np.uint8))
    px = DE.DiffImageProcessor([8, 8]).preprocess(img)["pixel_values"][0]
    assert px.shape == (3, 8, 8) and float(px.min()) == 1.0
    twin = DE.DiffImageProcessor([8, 8]).device_twin("cpu")                            # the device-side twin is constructible on any host
    assert type(twin).__name__ == "DevicePreprocessor"


def test_delay_load_gives_config_only_tower():
    cfg = SimpleNamespace(mm_vision_tower='openai/clip-vit-large-patch14-336', mm_vision_select_layer=-2, mm_vision_select_feature='patch')
    t = B.build_vision_tower(cfg, delay_load=True)
    assert not t.is_loaded and t.hidden_size == 1024 and t.num_patches == 576
    t2 = B.build_siglip_vision_tower(SimpleNamespace(mm_vision_tower='google/siglip-base-patch16-224', mm_vision_select_layer=-2), delay_load=True)
    assert t2.select_feature == 'cls_patch' and t2.hidden_size == 768 and t2.num_patches == 196


def test_cfg_view_narrows_both_tower_names():
    base = SimpleNamespace(mm_vision_tower="openai/clip-vit-large-patch14.runwayml/stable-diffusion-v1-5", vision_tower="whatever", t=261, img_size=768)
    v = LA._CfgView(base, "runwayml/stable-diffusion-v1-5")
    assert v.mm_vision_tower == v.vision_tower == "runwayml/stable-diffusion-v1-5" and v.t == 261 and v.img_size == 768
    assert LA.VisionEncoderStack._split(base.mm_vision_tower) == ["openai/clip-vit-large-patch14", "runwayml/stable-diffusion-v1-5"]


def test_fusion_id_splitting():
    s = LA.VisionEncoderStack._split
    assert s('openai/clip-vit-large-patch14-336.facebook/dinov2-large') == ['openai/clip-vit-large-patch14-336', 'facebook/dinov2-large']
    assert s('openai/clip-vit-large-patch14') == ['openai/clip-vit-large-patch14']


def test_projector_factory_types_and_state_dict_names():
    p = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=128, hidden_size=256))
    assert sorted(p.state_dict()) == ['0.bias', '0.weight', '2.bias', '2.weight']        # what llava_arch.py:183-189 loads
    assert p.state_dict()['2.weight'].shape == (256, 256)
    assert sorted(build_vision_projector(SimpleNamespace(mm_projector_type='linear', mm_hidden_size=128, hidden_size=256)).state_dict()) == ['0.bias', '0.weight']
    with pytest.raises(ValueError, match="Unknown projector type"):
        build_vision_projector(SimpleNamespace(mm_projector_type='bogus', mm_hidden_size=1, hidden_size=1))


def test_specs_and_interpolation():
    s = VW.SPECS['facebook/dinov2-large']
    assert s.tokens == 257 and s.pos_grid == 37
    pos = torch.randn(1 + 37 * 37, 8)
    out = VW.interpolate_pos(pos, True, 16)
    assert out.shape == (257, 8) and torch.equal(out[0], pos[0])
    assert VW.interpolate_pos(pos, True, 37) is pos
    w = VW.synthetic_weights(VW.tiny_spec("siglip", d=128, heads=2, mlp=256), seed=3)
    assert w["cls"] is None and w["patch_b"] is not None and len(w["layers"]) == 3
    w2 = VW.synthetic_weights(VW.tiny_spec("siglip", d=128, heads=2, mlp=256), seed=3)
    assert torch.equal(w["layers"][1]["wqkv"], w2["layers"][1]["wqkv"])                    # RandomState stream is stable
    # a change of resolution always resizes from the checkpoint's own grid, once (HF interpolate_pos_encoding: 37 -> 24), never from an
    # already-resized table (37 -> 16 -> 24); ADVICE r2
    native = VW.SPECS['facebook/dinov2-large'].at_resolution(37 * 14)
    w0 = {"pos": pos, "layers": []}
    s224, w224 = VW.weights_at_resolution(native, w0, 224)
    s336, w336 = VW.weights_at_resolution(s224, w224, 336)
    assert w224["pos"].shape[0] == 257 and w336["pos"].shape[0] == 577
    assert torch.equal(w336["pos"], VW.interpolate_pos(pos, True, 24))
    assert not torch.allclose(w336["pos"], VW.interpolate_pos(VW.interpolate_pos(pos, True, 16), True, 24), atol=1e-3)
    flat = VW.flatten(w)
    back = VW.unflatten(flat)
    assert torch.equal(back["layers"][2]["w2"], w["layers"][2]["w2"]) and back["cls"] is None


def test_extract_feature_diffusion_dispatch_shapes(tmp_path, monkeypatch):
    """extract_feature.py:68-103 post-processing per `feature`, with a stand-in featurizer (no device work)."""
    import numpy as np
    import torch
    from PIL import Image
    from law_of_vision_representation_in_mllms_amd.C_score import extract_feature as EF
    assert {k: v[2] for k, v in EF._DIFT.items()} == {"DIFT2.1": 768, "DIFT1.5": 768, "DIFTXL": 512, "IMDIFT": 768, "DiTDIFT": 512, "SD3DIFT": 512}
    calls = []

    class Stub:
        def forward(self, px, prompt, ensemble_size, post_noise=None, ddim_noise=None):
            calls.append((tuple(px.shape), px.dtype, prompt, ensemble_size))
            B = px.shape[0]
            f = torch.arange(B * 6 * 2 * 3, dtype=torch.float32).view(B, 6, 2, 3)
            return f.squeeze() if B == 1 else f
    src = tmp_path / "JPEGImages" / "cat"
    src.mkdir(parents=True)
    for i in range(3):
        Image.fromarray(np.full((20, 30, 3), 40 * i, np.uint8)).save(src / f"im{i}.jpg")
    monkeypatch.setattr(EF, "_state", SimpleNamespace(dift=Stub(), img_size=32, suffix="dift1.5", batch=2, kind="sd"))
    one = EF.extract_features(str(src / "im0.jpg"))
    assert one.shape == (1, 6, 2, 3) and calls[-1] == ((1, 3, 32, 32), torch.bfloat16, '', 1)
    EF.process_images(str(tmp_path / "JPEGImages"), str(tmp_path / "out"))
    f = torch.load(tmp_path / "out" / "cat" / "im2_dift1.5.pt")
    assert f.shape == (1, 6, 2, 3)
    EF._state.kind = "dit"                                    # DiT maps are stored transposed (extract_feature.py:95)
    t = EF.extract_features(str(src / "im0.jpg"))
    assert t.shape == (1, 6, 3, 2) and torch.equal(t, one.permute(0, 1, 3, 2))
    with pytest.raises(KeyError):
        EF.configure("NOPE")


def test_extract_feature_prefetch_keeps_order_and_pairs(monkeypatch):
    """The decode pool hands batches back in walk order whatever the worker count (files must match the serial run)."""
    import time
    import torch
    from law_of_vision_representation_in_mllms_amd.C_score import extract_feature as EF
    monkeypatch.setattr(EF, "_state", SimpleNamespace(img_size=4, batch=3))

    def load(path, size):
        time.sleep(0.002 * (7 - path % 7))                   # later items finish first
        return torch.full((2,), float(path))
    chunks = [[(i, f"o{i}") for i in range(s, min(s + 3, 10))] for s in range(0, 10, 3)]
    for workers in (1, 4):
        got = list(EF._prefetched(chunks, load, workers))
        assert [c for c, _ in got] == chunks
        assert [px[:, 0].tolist() for _, px in got] == [[float(i) for i, _ in c] for c in chunks]
    assert list(EF._prefetched([], load, 4)) == []
