"""Full-size parity of every ViT family of the paper's settings, of the LLaVA-1.5 projector at its real widths and of the batch-256
headline shape, against the fp32 CPU oracle (VERDICT r1 "parity is green but narrow at full size").  bf16 engine: the admissible error
is tied to the error of the SAME oracle run in bf16 (what the reference does on a GPU: model.to(bfloat16)), as in
test_gpu_kernels.py::test_vit_l14_336_full_size_parity.  The fp32 engine's full-size checks live in test_gpu_f32.py."""
import os

import numpy as np
import pytest
import torch

from law_of_vision_representation_in_mllms_amd import _lib, engine
from law_of_vision_representation_in_mllms_amd import vit_weights as VW
from oracle import projector as OP, vit as OV

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def fast_weights(spec, seed, n_layers):
    """random-init weights drawn with torch's generator (seconds instead of the numpy stream's tens of seconds at 300 M parameters);
    the SAME dict feeds the engine and the oracle, so the stream does not need to be version-stable here."""
    os.environ["VISREP_FAST_SYNTHETIC"] = "1"
    try:
        return VW.synthetic_weights(spec, seed=seed, n_layers=n_layers)
    finally:
        os.environ.pop("VISREP_FAST_SYNTHETIC", None)


CASES = {
    # id: (registry id, input side, layers run = hidden_states[-2], images, select_feature)
    "clip_l14_224": ("openai/clip-vit-large-patch14", 224, 23, 2, "patch"),                    # BASELINE configs[0]'s tower
    "openclip_l14_224_gelu": ("laion/CLIP-ViT-L-14-laion2B-s32B-b82K", 224, 23, 2, "patch"),
    "dinov2_l_224": ("facebook/dinov2-large", 224, 23, 2, "patch"),                            # BASELINE configs[3]'s tower
    "dinov2_l_336": ("facebook/dinov2-large", 336, 23, 2, "patch"),                            # CLIP336+DINOv2 fusion / C score at 336
    "siglip_b16_224": ("google/siglip-base-patch16-224", 224, 11, 3, "cls_patch"),
}


@pytest.mark.parametrize("case", list(CASES))
def test_vit_family_full_size_parity(case):
    name, side, n_layers, n_img, sel = CASES[case]
    base = VW.SPECS[name]
    native = base.at_resolution(base.pos_grid * base.patch) if base.pos_grid else base
    w0 = fast_weights(native, 3, n_layers)
    spec, w = VW.weights_at_resolution(native, w0, side)                          # DINOv2: bicubic position-embedding resize 37 -> 16 / 24
    rs = np.random.RandomState(11)
    px = bf(torch.from_numpy(rs.standard_normal((n_img, 3, side, side)).astype(np.float32)))
    hid = engine.VitEngine(spec, w, DEV).forward(px.to(DEV), n_layers=n_layers)
    got = hid if sel == "cls_patch" else hid[:, 1:]
    want = OV.tower_features(spec, w, px.float(), select_layer=n_layers, select_feature=sel)
    ref_bf16 = OV.tower_features(spec, w, px.float(), select_layer=n_layers, select_feature=sel, dtype=torch.bfloat16)
    assert got.shape == want.shape and torch.isfinite(got.float()).all()
    e_hip, e_ref = rel(got, want), rel(ref_bf16, want)
    assert e_hip < max(1.5 * e_ref, 2e-2), (case, e_hip, e_ref)


@pytest.mark.parametrize("width", [1024, 2048])
def test_projector_at_llava_widths(width):
    """mlp2x_gelu at LLaVA-1.5's real widths: 1024 -> 4096 -> 4096 (one ViT-L tower) and 2048 -> 4096 -> 4096 ('.'-fusion of two),
    on one image's 576 tokens + a ragged tail; bf16 MFMA path vs the fp32 oracle, and the fp32 path to fp32 rounding."""
    from types import SimpleNamespace
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_projector.builder import build_vision_projector
    torch.manual_seed(width)
    p = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=width, hidden_size=4096))
    x = torch.randn(2, 577, width) * 1.5
    want = OP.mlp_gelu(x, [p[0].weight.detach(), p[2].weight.detach()], [p[0].bias.detach(), p[2].bias.detach()])
    assert rel(p(x.to(DEV)), want) < 2e-5                                           # fp32 parameters: exact-fp32 MFMA path
    pb = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=width, hidden_size=4096))
    pb.load_state_dict(p.state_dict())
    got = pb.to(torch.bfloat16)(bf(x).to(DEV))
    ref_bf16 = OP.mlp_gelu(bf(x).float(), [bf(p[0].weight.detach()).float(), bf(p[2].weight.detach()).float()], [p[0].bias.detach(), p[2].bias.detach()])
    assert got.dtype == torch.bfloat16 and rel(got, ref_bf16) < 1e-2


def test_headline_batch_256_strided_sample_vs_oracle():
    """BASELINE configs[1] exactly as bench.py runs it (CLIP ViT-L/14-336, batch 256, bf16, 23 layers): images 0, 85, 170 and 255 of the
    batch against the fp32 oracle - row tiles of the 256 x 577-row GEMMs, the 128x128 tail launches and the attention's global key
    tiles all sit differently for them - and the same images run alone give the same features (batch invariance)."""
    spec = VW.SPECS["openai/clip-vit-large-patch14-336"]
    w = fast_weights(spec, 1, 23)
    g = torch.Generator().manual_seed(2)
    px = bf(torch.randn(256, 3, 336, 336, generator=g))
    eng = engine.VitEngine(spec, w, DEV)
    out = eng.forward(px.to(DEV), n_layers=23)[:, 1:]
    assert out.shape == (256, 576, 1024) and torch.isfinite(out.float()).all()
    ids = [0, 85, 170, 255]
    want = OV.tower_features(spec, w, px[ids].float(), select_layer=23, select_feature="patch")
    ref_bf16 = OV.tower_features(spec, w, px[ids[:1]].float(), select_layer=23, select_feature="patch", dtype=torch.bfloat16)
    e_ref = rel(ref_bf16, want[:1])
    for j, i in enumerate(ids):
        assert rel(out[i], want[j]) < max(1.5 * e_ref, 2e-2), (i, rel(out[i], want[j]), e_ref)
    alone = eng.forward(px[ids].to(DEV), n_layers=23)[:, 1:]
    assert rel(alone, out[ids]) < 2e-2                                          # two bf16 runs with different tilings: each ~1.2e-2 from the oracle
