"""GPU tests of the drop-in surface: tower classes, '.'-fusion + mm_projector (encode_images), feature dump, A_score.compute
and C_score.extract_feature / pck_train on device — each against the CPU oracle on the same inputs."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(__file__))
from test_host_cscore import G, make_tree, eval_args, eval_args_two  # noqa: E402

from law_of_vision_representation_in_mllms_amd import vit_weights as VW  # noqa: E402
from law_of_vision_representation_in_mllms_amd.A_score import compute as AC  # noqa: E402
from law_of_vision_representation_in_mllms_amd.C_score import extract_feature as EF  # noqa: E402
from law_of_vision_representation_in_mllms_amd.C_score import pck_train as PT  # noqa: E402
from law_of_vision_representation_in_mllms_amd.llava.model import llava_arch as LA  # noqa: E402
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder import _vit_tower as VT  # noqa: E402
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_projector.builder import build_vision_projector  # noqa: E402
from oracle import ascore as OA, projector as OP, vit as OV  # noqa: E402

DEV = "cuda:0"

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture
def small_towers(monkeypatch):
    """Registry ids keep their names but get small head_dim-64 specs so the test runs in seconds."""
    specs = {
        'openai/clip-vit-large-patch14': VW.tiny_spec("clip", image_size=42, patch=14, d=128, heads=2, mlp=256, layers=3),
        'openai/clip-vit-large-patch14-336': VW.tiny_spec("clip", image_size=56, patch=14, d=128, heads=2, mlp=256, layers=3),
        'facebook/dinov2-large': VW.tiny_spec("dinov2", image_size=42, patch=14, d=128, heads=2, mlp=256, layers=3),
        'google/siglip-base-patch16-224': VW.tiny_spec("siglip", image_size=48, patch=16, d=128, heads=2, mlp=256, layers=3),
    }
    monkeypatch.setattr(VW, "SPECS", {**VW.SPECS, **specs})
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    return specs


def tower_oracle(spec, px, select_feature):
    w = VW.synthetic_weights(spec, seed=1)
    return OV.tower_features(spec, w, px, -2, select_feature)


def test_tower_classes_match_oracle(small_towers):
    for name, builder_key, sel in [('openai/clip-vit-large-patch14', 'openai/clip-vit-large-patch14', 'patch'),
                                   ('facebook/dinov2-large', 'facebook/dinov2-large', 'patch'),
                                   ('google/siglip-base-patch16-224', 'google/siglip-base-patch16-224', 'cls_patch')]:
        cfg = SimpleNamespace(mm_vision_tower=name, mm_vision_select_layer=-2, mm_vision_select_feature='patch')
        tower = LA.build_function_mapping[builder_key](cfg)
        spec = small_towers[name]
        px = torch.randn(3, 3, spec.image_size, spec.image_size)
        out = tower(px)
        assert out.dtype == px.dtype and out.shape == (3, spec.tokens - (1 if sel == 'patch' else 0), spec.d)
        assert rel(out, tower_oracle(spec, px, sel)) < 2e-2
        lst = tower([px[0], px[1]])                                    # list-of-images path (clip_encoder.py:41-46)
        assert len(lst) == 2 and rel(lst[1][0], out[1]) < 1e-2
        assert tower.num_patches == spec.num_patches and tower.hidden_size == spec.d and tower.dummy_feature.shape == (1, spec.d)
    tower.select_feature = "bogus"
    with pytest.raises(ValueError, match="Unexpected select feature"):
        tower(px)


def test_projector_matches_reference_golden():
    # golden widths (48 -> 96) are not bf16-MFMA-tileable: in bf16 the factory must refuse them loudly, not fall back
    p = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=48, hidden_size=96)).to(torch.bfloat16)
    with pytest.raises(ValueError, match="multiples of 64"):
        p(torch.zeros(2, 48).cuda())
    p = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=128, hidden_size=256))
    x = torch.randn(2, 10, 128)
    want = OP.mlp_gelu(x, [p[0].weight, p[2].weight], [p[0].bias, p[2].bias])
    assert rel(p(x.cuda()), want) < 1e-5                                 # fp32 parameters: the exact-fp32 path
    pb = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=128, hidden_size=256))
    pb.load_state_dict(p.state_dict())
    assert rel(pb.to(torch.bfloat16)(x.cuda()), want) < 1e-2              # bf16 parameters (LLaVA's model.to(bfloat16)): bf16 MFMA path
    # diffusion-tower widths (SD 1280, DiT 4608, SD3 6144 -> LLM width): only multiples of 64 are required
    p = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=1280, hidden_size=192)).to(torch.bfloat16)
    x = torch.randn(3, 5, 1280)
    assert rel(p(x.cuda()), OP.mlp_gelu(x, [p[0].weight.float(), p[2].weight.float()], [p[0].bias.float(), p[2].bias.float()])) < 1e-2


def test_fusion_stack_encode_images_and_feature_dump(small_towers, tmp_path):
    cfg = SimpleNamespace(mm_vision_tower='openai/clip-vit-large-patch14.facebook/dinov2-large', mm_vision_select_layer=-2,
                          mm_vision_select_feature='patch', mm_projector_type='mlp2x_gelu', hidden_size=256)
    stack = LA.VisionEncoderStack(cfg)
    assert cfg.mm_hidden_size == 256
    px = torch.randn(2, 3, 42, 42)
    feats = stack.encode_images([px, px])                               # one tensor per tower (llava_arch.py:278-285)
    assert feats.shape == (2, 9, 256)
    f_cat = torch.cat([tower_oracle(small_towers['openai/clip-vit-large-patch14'], px, 'patch'),
                       tower_oracle(small_towers['facebook/dinov2-large'], px, 'patch')], dim=-1)
    want = OP.mlp_gelu(f_cat, [stack.mm_projector[0].weight, stack.mm_projector[2].weight],
                       [stack.mm_projector[0].bias, stack.mm_projector[2].bias])
    assert rel(feats, want) < 3e-2
    for i in range(3):
        full = LA.save_tensor_to_folder(feats[0].cpu(), str(tmp_path / "dump"), max_tensors=3, exit_when_full=False)
    assert full and sorted(os.listdir(tmp_path / "dump")) == ["tensor_1.pt", "tensor_2.pt", "tensor_3.pt"]


def test_ascore_compute_on_device(tmp_path):
    rs = np.random.RandomState(4)
    data = {}
    for sub, nt in dict(clip336=40, clip224=24, encA=33).items():
        os.makedirs(tmp_path / sub)
        data[sub] = []
        for i in range(1, 6):
            t = torch.from_numpy(rs.standard_normal((nt, 256)).astype(np.float32)).to(torch.bfloat16)
            torch.save(t, tmp_path / sub / f"tensor_{i}.pt")
            data[sub].append(t)
    res = AC.compute(str(tmp_path), ["clip336", "encA"], n_images=5, verbose=False)
    for enc in res:
        want, _, _ = OA.a_score(data[enc], data["clip336"], data["clip224"])
        assert abs(res[enc] - want) < 1e-4 * abs(want)
    assert abs(res["clip336"] - 0.5 * (1 + OA.a_score(data["clip336"], data["clip224"], data["clip224"])[1])) < 1e-4


def test_extract_feature_and_pck_train_on_device(small_towers, tmp_path):
    # extract_feature: JPEG -> resize -> (x/255-.5)*2 -> tower -> [1, C, g, g] files named <img>_<suffix>.pt
    src = tmp_path / "JPEGImages" / "cat"
    os.makedirs(src)
    rs = np.random.RandomState(6)
    for i in range(3):
        Image.fromarray(rs.randint(0, 255, (50 + i, 70, 3), dtype=np.uint8)).save(src / f"im{i}.jpg")
    EF.configure("DINOv2", img_size=42, suffix="dino", batch=2)
    EF.process_images(str(tmp_path / "JPEGImages"), str(tmp_path / "features"))
    f = torch.load(tmp_path / "features" / "cat" / "im1_dino.pt")
    assert f.shape == (1, 128, 3, 3)
    one = EF.extract_features(str(src / "im1.jpg"))
    assert rel(one, f) < 1e-2
    # resize + normalise on the device (decode on the pool): bit-identical pixels, so bit-identical files
    EF._state.device_preprocess = True
    try:
        EF.process_images(str(tmp_path / "JPEGImages"), str(tmp_path / "features_dev"), workers=4)
    finally:
        EF._state.device_preprocess = False
    for i in range(3):
        assert torch.equal(torch.load(tmp_path / "features_dev" / "cat" / f"im{i}_dino.pt"), torch.load(tmp_path / "features" / "cat" / f"im{i}_dino.pt"))
    # ADAPT_FLIP's mirrored feature files: the same extraction on the left-right flipped image
    EF.process_images(str(tmp_path / "JPEGImages"), str(tmp_path / "features"), flip=True)
    ff = torch.load(tmp_path / "features" / "cat" / "im1_dino_flip.pt")
    pxf = EF._load_pixels(str(src / "im1.jpg"), 42).flip(-1).unsqueeze(0)       # mirroring commutes with the square resize only approximately
    assert ff.shape == f.shape and not torch.equal(ff, f)
    spec = small_towers['facebook/dinov2-large']
    px = EF._load_pixels(str(src / "im1.jpg"), 42).unsqueeze(0)
    want = tower_oracle(spec, px, 'patch').permute(0, 2, 1).reshape(1, 128, 3, 3)
    assert rel(f, want) < 2e-2
    # pck_train.eval on the mini SPair tree, device kernels this time, against the reference's eval() result
    root, z = make_tree(str(tmp_path / "spair"))
    p10, p05, p01, results = PT.eval(eval_args(root, 16), PT.DummyAggregationNetwork(), str(tmp_path), split="test")
    np.testing.assert_allclose([p10, p05, p01], z["eval.pck"], atol=1e-7)
    np.testing.assert_allclose(np.stack([r["src_kpts_pred"] for r in results]), z["eval.pred"], atol=5e-3)
    # geometry-aware metrics (COMPUTE_GEOAWARE_METRICS): per-category geo_score of the reference's run
    ga = eval_args(root, 16)
    ga.COMPUTE_GEOAWARE_METRICS = True
    scores = []
    spy = lambda *a, **k: (lambda r: (scores.append(r[1]), r)[1])(PT.compute_pck(*a, **k))
    g = PT.eval(ga, PT.DummyAggregationNetwork(), str(tmp_path), split="test", _compute=spy)
    np.testing.assert_allclose(g[:3], z["geo.pck"], atol=1e-7)
    np.testing.assert_allclose(np.array(scores, np.float64), z["geo.scores"], rtol=0, atol=1e-12)
    # two-encoder evaluator (pck_train_two.py) on the same tree, against the reference's own eval()
    from law_of_vision_representation_in_mllms_amd.C_score import pck_train_two as PT2
    q10, q05, q01, results2 = PT2.eval(eval_args_two(root, 16), PT2.DummyAggregationNetwork(), str(tmp_path), split="test")
    np.testing.assert_allclose([q10, q05, q01], z["eval2.pck"], atol=1e-7)
    np.testing.assert_allclose(np.stack([r["src_kpts_pred"] for r in results2]), z["eval2.pred"], atol=5e-3)


def test_extract_feature_diffusion_feature_on_device(tmp_path):
    """extract_feature with feature = "DIFT1.5" (full SD1.5 architecture, synthetic weights) at a small input side."""
    src = tmp_path / "JPEGImages" / "dog"
    os.makedirs(src)
    rs = np.random.RandomState(3)
    for i in range(3):
        Image.fromarray(rs.randint(0, 255, (60, 80, 3), dtype=np.uint8)).save(src / f"im{i}.jpg")
    EF.configure("DIFT1.5", img_size=128, synthetic_weights=True, batch=2)
    try:
        EF.process_images(str(tmp_path / "JPEGImages"), str(tmp_path / "features"))
        f = torch.load(tmp_path / "features" / "dog" / "im1_dift1.5.pt")
        assert f.shape == (1, 1280, 4, 4) and torch.isfinite(f.float()).all()
        assert EF.extract_features(str(src / "im1.jpg")).shape == (1, 1280, 4, 4)
    finally:
        EF._state.dift, EF._state.kind = None, "vit"


@pytest.mark.parametrize("w,h,ow,oh", [(640, 480, 224, 224), (500, 375, 448, 336), (100, 80, 224, 224), (768, 768, 224, 224), (17, 9, 5, 31),
                                       (224, 300, 224, 224)])
def test_device_resize_is_bit_exact_with_pil(w, h, ow, oh):
    from law_of_vision_representation_in_mllms_amd import device_preprocess as DP
    rs = np.random.RandomState(w * 7 + h)
    a = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 3] = rs.randint(0, 2, (h // 3, w, 3)) * 255
    want = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))
    got = DP.resize_u8(torch.from_numpy(a).to(DEV), (ow, oh)).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("pad,flip", [(False, False), (True, False), (False, True), (True, True)])
def test_batched_preprocess_equals_pil_per_image(pad, flip):
    """visrep_preprocess_u8_batch: one descriptor per image, two launches per batch - images of different sizes (down- and upscaled, one
    axis already at the target size, already square), optional expand2square canvas and mirror, resize + centre crop + normalise: equal to
    PIL + the numpy expression per image, and to the per-image device route, bit for bit."""
    from law_of_vision_representation_in_mllms_amd import device_preprocess as DP
    from law_of_vision_representation_in_mllms_amd.llava.mm_utils import expand2square
    rs = np.random.RandomState(3 + pad + 2 * flip)
    shapes = [(640, 427), (333, 500), (224, 224), (100, 80), (224, 300), (300, 224), (17, 9), (768, 768), (501, 224)]
    arrs = [rs.randint(0, 256, (h, w, 3), dtype=np.uint8) for w, h in shapes]
    bg = (122, 116, 104)
    mean, std = (0.481, 0.457, 0.408), (0.268, 0.261, 0.275)
    S, crop = 224, 196
    sizes, boxes, want = [], [], []
    for a in arrs:
        im = Image.fromarray(a)
        if flip:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        if pad:
            im = expand2square(im, bg)
        w, h = im.size
        ow, oh = (S, max(S, int(h * S / w))) if w <= h else (max(S, int(w * S / h)), S)      # shortest edge -> S
        l, t = (ow - crop) // 2, (oh - crop) // 2
        sizes.append((ow, oh))
        boxes.append((l, t, crop, crop))
        r = np.asarray(im.resize((ow, oh), Image.BICUBIC))[t: t + crop, l: l + crop].transpose(2, 0, 1).astype(np.float32)
        want.append((r / np.float32(255.0) - np.array(mean, np.float32)[:, None, None]) / np.array(std, np.float32)[:, None, None])
    dev = [torch.from_numpy(a).to(DEV) for a in arrs]
    got = DP.preprocess_batch(dev, sizes, boxes, mean, std, torch.float32, bg if pad else None, flip)
    assert got.shape == (len(arrs), 3, crop, crop)
    for i, w_ in enumerate(want):
        assert np.array_equal(got[i].cpu().numpy(), w_), (i, shapes[i])
    # the per-image route (three launches per image) agrees
    for i, d in enumerate(dev):
        x = d.flip(1) if flip else d
        if pad:
            x = DP.expand2square_u8(x, bg)
        assert torch.equal(DP.to_tensor(DP.resize_u8(x, sizes[i]), boxes[i], mean, std), got[i]), i
    # bf16 output + caller-provided output tensor; an empty batch
    out = torch.empty(len(arrs), 3, crop, crop, dtype=torch.bfloat16, device=DEV)
    assert DP.preprocess_batch(dev, sizes, boxes, mean, std, pad_background=bg if pad else None, flip=flip, out=out) is out
    assert torch.equal(out, got.to(torch.bfloat16))
    with pytest.raises(ValueError):
        DP.preprocess_batch(dev[:2], sizes[:2], [boxes[0], (0, 0, crop, crop - 1)], mean, std)


def test_device_preprocessor_equals_cpu_processors(tmp_path):
    """CLIP / DINOv2 / SigLIP processor geometry and arithmetic on the device: bit-identical fp32 pixel tensors; and the C-path
    loader of extract_feature."""
    from law_of_vision_representation_in_mllms_amd import device_preprocess as DP
    from law_of_vision_representation_in_mllms_amd import vit_weights as VW
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.image_processing import default_image_processor
    rs = np.random.RandomState(11)
    imgs = [Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)) for w, h in ((640, 427), (333, 500), (224, 224))]
    for name in ("openai/clip-vit-large-patch14-336", "facebook/dinov2-large", "google/siglip-base-patch16-224"):
        cpu = default_image_processor(VW.SPECS[name])
        want = cpu.preprocess(imgs)["pixel_values"]
        got = DP.DevicePreprocessor.like(cpu, DEV).preprocess(imgs)["pixel_values"]
        assert got.shape == want.shape and torch.equal(got.cpu(), want), name
    # llava/mm_utils.py process_images with image_aspect_ratio = 'pad' accepts the device processor as is
    from law_of_vision_representation_in_mllms_amd.llava import mm_utils as MU
    cpu = default_image_processor(VW.SPECS["openai/clip-vit-large-patch14-336"])
    cfg = SimpleNamespace(image_aspect_ratio="pad")
    assert torch.equal(MU.process_images(imgs, DP.DevicePreprocessor.like(cpu, DEV), cfg).cpu(), MU.process_images(imgs, cpu, cfg))
    imgs[0].save(tmp_path / "a.jpg")
    assert torch.equal(EF._load_pixels_device(str(tmp_path / "a.jpg"), 224, DEV).cpu(), EF._load_pixels(str(tmp_path / "a.jpg"), 224))
    # the diffusion towers' resize-only processor (diffusion_encoder.py:30-41: PIL resize to img_size, (x / 255 - 0.5) * 2) has a device twin too
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM.diffusion_encoder import DiffImageProcessor
    for side in (64, 96):
        dp = DiffImageProcessor([side, side])
        want = torch.stack([dp.preprocess(im)["pixel_values"][0] for im in imgs])
        got = dp.device_twin(DEV).preprocess(imgs)["pixel_values"]
        assert got.shape == want.shape == (3, 3, side, side) and torch.equal(got.cpu(), want), side
    with pytest.raises(ValueError, match="positive square"):
        DiffImageProcessor([0, 0]).device_twin(DEV)


def test_fused_pipelines_equal_the_file_route(small_towers, tmp_path):
    """pipeline.py: features stay in HBM; the scores equal the dump-to-disk route exactly."""
    from law_of_vision_representation_in_mllms_amd import pipeline as PL
    rs = np.random.RandomState(21)
    images = [Image.fromarray(rs.randint(0, 256, (60 + 3 * i, 80, 3), dtype=np.uint8)) for i in range(6)]
    # ---- A score: three encoder stacks (tower + mlp2x_gelu projector), file route = save_tensor_to_folder + A_score.compute
    stacks = {}
    for name, tower_id in dict(clip336='openai/clip-vit-large-patch14-336', clip224='openai/clip-vit-large-patch14', dino='facebook/dinov2-large').items():
        cfg = SimpleNamespace(mm_vision_tower=tower_id, mm_vision_select_layer=-2, mm_vision_select_feature='patch', mm_projector_type='mlp2x_gelu',
                              hidden_size=256)
        torch.manual_seed(len(name))
        stacks[name] = LA.VisionEncoderStack(cfg)
    fused = PL.a_scores_from_stacks(stacks, images, batch=4, verbose=False)
    for name, stack in stacks.items():
        proc = stack.get_vision_tower().image_processor
        for k, im in enumerate(images, 1):
            f = stack.encode_images(proc.preprocess([im])["pixel_values"])[0]
            os.makedirs(tmp_path / "bench" / name, exist_ok=True)
            torch.save(f.cpu(), tmp_path / "bench" / name / f"tensor_{k}.pt")
    filed = AC.compute(str(tmp_path / "bench"), list(stacks), n_images=len(images), verbose=False)
    for name in stacks:
        assert abs(fused[name] - filed[name]) < 2e-3 * abs(filed[name]), name          # batch-1 vs batched tower rounding only
    assert abs(fused["clip336"] - 0.5) > 1e-3 and fused.keys() == filed.keys()
    # ---- C score: dense maps straight from a tower vs extract_feature files + pck_train.eval
    root, z = make_tree(str(tmp_path / "spair"))
    jpeg = os.path.join(root, "JPEGImages")
    for cat in ("aeroplane", "cat"):
        os.makedirs(os.path.join(jpeg, cat), exist_ok=True)
        for i in range(5):
            Image.fromarray(rs.randint(0, 256, (70, 90, 3), dtype=np.uint8)).save(os.path.join(jpeg, cat, f"img{i}.jpg"))
    EF.configure("DINOv2", img_size=42, suffix="dinofile", batch=4)
    EF.process_images(jpeg, os.path.join(root, "features"))
    a = eval_args(root, 3)
    a.MODEL = "dinofile"
    want = PT.eval(a, PT.DummyAggregationNetwork(), str(tmp_path), split="test")

    def extract(paths):
        return EF._state.dift.forward(torch.stack([EF._load_pixels(p, 42) for p in paths]))
    got = PL.c_score_from_tower(a, extract, str(tmp_path), batch=3)
    np.testing.assert_allclose(got[:3], want[:3], atol=1e-7)
    np.testing.assert_allclose(np.stack([r["src_kpts_pred"] for r in got[3]]), np.stack([r["src_kpts_pred"] for r in want[3]]), atol=0.5)


def test_feature_dump_with_a_registry_tower(small_towers, tmp_path):
    """llava/feature/extract.py end to end on the device: registry -> DINOv2 tower -> one [N, C] bf16 file per image."""
    import json
    from law_of_vision_representation_in_mllms_amd.llava.feature import extract as FX
    rs = np.random.RandomState(8)
    os.makedirs(tmp_path / "imgs" / "coco")
    entries = []
    for i in range(3):
        Image.fromarray(rs.randint(0, 255, (90 + 10 * i, 120, 3), dtype=np.uint8)).save(tmp_path / "imgs" / "coco" / f"im{i}.jpg")
        entries.append({"image": f"coco/im{i}.jpg"})
    with open(tmp_path / "d.json", "w") as f:
        json.dump(entries, f)
    a = SimpleNamespace(vision_tower='facebook/dinov2-large', mm_vision_select_layer=-2, mm_vision_select_feature='patch', img_size=768,
                        data_path=str(tmp_path / "d.json"), image_folder=str(tmp_path / "imgs"), image_aspect_ratio='pad',
                        per_device_train_batch_size=2, feature_dir=str(tmp_path / "feats"))
    assert FX.inference(a, a, a) == 3
    spec = small_towers['facebook/dinov2-large']                   # img_size = 768 is the diffusion towers' field: ignored here
    for i in range(3):
        got = torch.load(tmp_path / "feats" / "coco" / f"im{i}.pt")
        px = FX.load_image(str(tmp_path / "imgs" / "coco" / f"im{i}.jpg"), a.image_processor, 'pad')[None].to(torch.bfloat16)
        assert got.shape == (9, 128) and got.dtype == torch.bfloat16
        assert rel(got, tower_oracle(spec, px.float(), 'patch')[0]) < 2e-2


def test_device_lanczos_and_geoaware_loader_are_bit_exact():
    """Device resampling with the LANCZOS / BILINEAR tables and the GeoAware-SC loader (LANCZOS + zero / edge padding) against outputs
    of PIL and of the reference's utils_correspondence.resize (tests/golden/georesize.npz)."""
    from law_of_vision_representation_in_mllms_amd import device_preprocess as DP
    z = np.load(os.path.join(G, "georesize.npz"))
    T = int(z["target"])
    for tag in ("land", "port", "square", "wide", "up"):
        a = torch.from_numpy(z[f"{tag}.in"]).to(DEV)
        assert np.array_equal(DP.resize_u8(a, (37, 29), "lanczos").cpu().numpy(), z[f"{tag}.lanczos"]), tag
        assert np.array_equal(DP.resize_u8(a, (37, 29), "bilinear").cpu().numpy(), z[f"{tag}.bilinear"]), tag
        for edge in (False, True):
            got = DP.geoaware_resize(a, T, edge)
            assert got.shape == (T, T, 3) and np.array_equal(got.cpu().numpy(), z[f"{tag}.edge{int(edge)}"]), (tag, edge)


# ------------------------------------------------------------------------------------------------ ADAPT_FLIP on the device (§8f N4)
def test_mutual_nn_distance_matches_the_reference():
    """visrep_gram_pairs_f32 + visrep_mutual_nn_distance vs the reference's get_distance_mutual_nn (tests/golden/adaptflip.npz), the
    drop-in function, and a batch of pairs against the oracle."""
    from law_of_vision_representation_in_mllms_amd import cscore_ops
    from law_of_vision_representation_in_mllms_amd.C_score.utils import utils_correspondence as UC
    from oracle import cscore as OC
    zf = np.load(f"{G}/adaptflip.npz")
    for tag, P in (("d6", 6), ("d16", 16)):
        f1, f2 = torch.from_numpy(zf[f"{tag}.f1"]), torch.from_numpy(zf[f"{tag}.f2"])
        bank = torch.stack([f1, f2]).to(DEV)
        got = cscore_ops.mutual_nn_distance(bank, torch.tensor([0, 1]), torch.tensor([1, 0]), P).cpu()
        assert abs(got[0].item() - float(zf[f"{tag}.dist"])) < 1e-4, tag
        want10 = OC.mutual_nn_distance(OC.normalize_feats(f2[None]), OC.normalize_feats(f1[None])).item()
        assert abs(got[1].item() - want10) < 1e-4
        d = UC.get_distance_mutual_nn(OC.normalize_feats(f1[None]), OC.normalize_feats(f2[None]))
        assert abs(d.item() - float(zf[f"{tag}.dist"])) < 1e-4
    g = torch.Generator().manual_seed(9)
    bank = torch.randn(7, 24 * 24, 64, generator=g) + 2.0 * torch.randn(1, 24 * 24, 64, generator=g)      # P = 24: 9 columns per lane
    i1, i2 = torch.tensor([0, 3, 6, 2, 5]), torch.tensor([1, 3, 0, 4, 6])
    got = cscore_ops.mutual_nn_distance(bank.to(DEV), i1, i2, 24, chunk=2).cpu()                           # chunked launches
    for n, (a, b) in enumerate(zip(i1.tolist(), i2.tolist())):
        want = OC.mutual_nn_distance(OC.normalize_feats(bank[a][None]), OC.normalize_feats(bank[b][None])).item()
        assert abs(got[n].item() - want) < 1e-4, (n, got[n].item(), want)
    assert got[1].item() < 1e-3                                                                            # a map against itself
    # 48 x 48 maps (the 768-px diffusion towers) and 60 x 60 maps (GeoAware-SC's own grid): 36 / 57 columns per lane
    for P in (48, 60):
        bank = torch.randn(3, P * P, 32, generator=g) + 1.5 * torch.randn(1, P * P, 32, generator=g)
        got = cscore_ops.mutual_nn_distance(bank.to(DEV), torch.tensor([0, 2]), torch.tensor([1, 0]), P).cpu()
        for n, (a, b) in enumerate(((0, 1), (2, 0))):
            want = OC.mutual_nn_distance(OC.normalize_feats(bank[a][None]), OC.normalize_feats(bank[b][None])).item()
            assert abs(got[n].item() - want) < 1e-4, (P, n, got[n].item(), want)
    # the epsilon of normalize_feats reaches the kernel (it was hard-wired to 1e-10)
    bank = torch.randn(2, 16 * 16, 16, generator=g)
    got = cscore_ops.mutual_nn_distance(bank.to(DEV), torch.tensor([0]), torch.tensor([1]), 16, eps=0.5).cpu()
    want = OC.mutual_nn_distance(OC.normalize_feats(bank[0][None], 0.5), OC.normalize_feats(bank[1][None], 0.5)).item()
    assert abs(got[0].item() - want) < 1e-4, (got[0].item(), want)


def test_patch_index_on_the_far_border_raises_like_the_reference():
    """A key point at x = y = ANNO_SIZE maps to flat patch index P*P + P - 1 >= P*P: IndexError in the reference's tensor indexing
    (utils_correspondence.py:360); the transfer kernel indexes unclamped, so compute_pck refuses before the launch."""
    from law_of_vision_representation_in_mllms_amd.C_score import pck_train as PT
    P, C_ = 6, 8
    args = SimpleNamespace(NUM_PATCHES=P, SOFT_EVAL=True, SOFT_EVAL_WINDOW=2, ANNO_SIZE=840, EVAL_DATASET='spair', KPT_RESULT=False, ENSEMBLE=1,
                           MODEL="m", ADAPT_FLIP=False, TOTAL_SAVE_RESULT=0, COMPUTE_GEOAWARE_METRICS=False)
    bank = torch.randn(2, P * P, C_, device=DEV)
    kps = torch.zeros(2, 3, 3)
    kps[:, :, 2] = 1
    kps[0, 1, :2] = 840.0                                                   # the source key point on the far corner
    files = ["a.jpg", "b.jpg"]
    with pytest.raises(IndexError, match="out of bounds"):
        PT._compute_pck(args, ".", PT.DummyAggregationNetwork(), files, kps, "cat", None, np.array([100.0]), (bank, np.array([0, 1], np.int32), 0, "pc"), models=("m",))
    kps[0, 1, :2] = 839.0
    PT._compute_pck(args, ".", PT.DummyAggregationNetwork(), files, kps, "cat", None, np.array([100.0]), (bank, np.array([0, 1], np.int32), 0, "pc"), models=("m",))


def test_adapt_flip_eval_on_device_matches_reference_eval(tmp_path):
    from test_host_cscore import make_flip_tree
    root, z, zf = make_flip_tree(str(tmp_path))
    a = eval_args(root, 16)
    a.ADAPT_FLIP, a.MUTUAL_NN = True, True
    p10, p05, p01, results = PT.eval(a, PT.DummyAggregationNetwork(), str(tmp_path), split="test")
    np.testing.assert_allclose([p10, p05, p01], zf["eval.pck"], atol=1e-7)
    np.testing.assert_allclose(np.stack([r["src_kpts_pred"] for r in results]), zf["eval.pred"], atol=5e-3)


def test_mask_distance_matches_the_reference():
    """ADAPT_FLIP without MUTUAL_NN: cscore_ops.masked_nn_distance / the drop-in get_distance (visrep_masked_nn_min_f32 behind torch's resize glue)
    against the reference's get_distance run as it stands on 60 x 60 maps (tests/golden/maskdist.npz, incl. the exact-zeros case whose answer is
    dominated by -100000 entries: difference-first arithmetic), then against the oracle on grids the reference cannot take (P = 16, 24; C not a
    multiple of 4; a one-cell source mask; an empty one)."""
    from law_of_vision_representation_in_mllms_amd import cscore_ops
    from law_of_vision_representation_in_mllms_amd.C_score.utils import utils_correspondence as UC
    from oracle import cscore as OC
    z = np.load(f"{G}/maskdist.npz")
    for tag in ("a", "b", "zeros"):
        f1, f2 = (torch.from_numpy(z[f"{tag}.{k}"].astype(np.float32))[None] for k in ("f1", "f2"))
        m1, m2 = (torch.from_numpy(z[f"{tag}.{k}"].astype(np.float32)) for k in ("m1", "m2"))
        want = float(z[f"{tag}.dist"])
        got = UC.get_distance(f1, f2, m1, m2)
        assert got.dim() == 0 and abs(got.item() - want) <= 2e-6 * abs(want), (tag, got.item(), want)
    g = torch.Generator().manual_seed(4)
    for P, C_ in ((16, 64), (24, 30), (16, 1024)):
        base = torch.randn(P * P, C_, generator=g)
        f1 = OC.normalize_feats((base + 0.4 * torch.randn(P * P, C_, generator=g))[None])
        f2 = OC.normalize_feats((base.roll(5, 0) + 0.4 * torch.randn(P * P, C_, generator=g))[None])
        m1, m2 = (torch.rand(40, 56, generator=g) > 0.6).float(), (torch.rand(33, 47, generator=g) > 0.3).float()
        want = OC.masked_nn_distance(f1, f2, m1, m2).item()
        got = cscore_ops.masked_nn_distance(f1.to(DEV), f2.to(DEV), m1, m2).item()
        assert abs(got - want) <= 2e-6 * abs(want), (P, C_, got, want)
        one = torch.zeros(64, 64)
        one[10, 20] = 1
        assert abs(cscore_ops.masked_nn_distance(f1.to(DEV), f2.to(DEV), one, m2).item() - OC.masked_nn_distance(f1, f2, one, m2).item()) < 1e-5
        assert torch.isnan(cscore_ops.masked_nn_distance(f1.to(DEV), f2.to(DEV), torch.zeros(8, 8), m2))
    with pytest.raises(AttributeError, match="unsqueeze"):
        cscore_ops.masked_nn_distance(f1.to(DEV), f2.to(DEV), None, m2)


def test_adapt_flip_with_the_mask_distance_on_device(tmp_path):
    """pck_train.eval with ADAPT_FLIP, MUTUAL_NN off, on the mini tree with `_mask.png` / `_mask_flip.png` files: the device route (transfer
    kernel + visrep_masked_nn_min_f32) equals the oracle chain pair by pair, and both branches of the flip decision are taken."""
    from test_host_cscore import make_flip_tree, masked_flip_expectation, write_masks
    from oracle import cscore as OC
    root, z, zf = make_flip_tree(str(tmp_path))
    cats = {"aeroplane": 4, "cat": 3}
    write_masks(root, cats)
    a = eval_args(root, 16)
    a.ADAPT_FLIP, a.MUTUAL_NN = True, False
    p10, p05, p01, results = PT.eval(a, PT.DummyAggregationNetwork(), str(tmp_path), split="test")
    want, flips = masked_flip_expectation(root, a, z, zf, cats, lambda *x: OC.masked_nn_distance(*x).item())
    assert 0 < flips < len(want)
    got = np.stack([r["src_kpts_pred"] for r in results])
    for n, (w, used) in enumerate(want):
        np.testing.assert_allclose(got[n][used.numpy()], w.numpy(), atol=5e-3, err_msg=str(n))


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_aggregation_network_on_device_matches_reference(tag):
    """GeoAware-SC's supervised post-processor (projection_network.py:15-125) on the exact-fp32 path vs the reference module's output."""
    from test_oracle_golden import load_aggnet_case
    from law_of_vision_representation_in_mllms_amd.C_score.model_utils.projection_network import AggregationNetwork
    cfg, sd, x, want = load_aggnet_case(tag)
    net = AggregationNetwork(device=DEV, **cfg)
    net.load_state_dict(sd)
    got = net(x.to(DEV))
    assert got.shape == want.shape and got.dtype == torch.float32
    torch.testing.assert_close(got.cpu(), want, rtol=2e-4, atol=2e-4)
    again = net(x)                                                              # CPU tensor in: moved to the device, cached packing reused
    assert torch.equal(again, got)


# ------------------------------------------------------------------------------------------------ towers built from checkpoint DIRECTORIES
@pytest.mark.parametrize("fmt", ["safetensors", "bin2"])
@pytest.mark.parametrize("kind", ["clip_full", "dinov2", "siglip"])
def test_tower_built_from_a_checkpoint_directory_equals_the_hf_model(tmp_path, kind, fmt):
    """The reference's `from_pretrained(path)` route end to end (clip_encoder.py:22-27): a tiny HF model saved with save_pretrained
    (safetensors, and a two-shard .bin) -> tower class pointed at the DIRECTORY -> features equal HF's own forward of that model (the
    arithmetic the reference runs) and the oracle on the packed weights.  tests/test_host_checkpoints.py pins the loader itself."""
    from test_host_checkpoints import tiny_hf_model, write_checkpoint
    model, Tower, family = tiny_hf_model(kind)
    path = str(tmp_path / kind)
    write_checkpoint(model, path, fmt, {"height": 42, "width": 42} if kind == "dinov2" else None)
    sel = "cls_patch" if family == "siglip" else "patch"
    tower = Tower(path, SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature=sel, device=DEV))
    assert tower.is_loaded and tower.vision_tower_name == path
    side = tower.spec.image_size
    px = torch.randn(3, 3, side, side, generator=torch.Generator().manual_seed(2))
    got = tower(px)
    with torch.no_grad():
        vm = model.vision_model if kind == "clip_full" else model
        kw = {"interpolate_pos_encoding": True} if kind == "dinov2" and "interpolate_pos_encoding" in vm.forward.__code__.co_varnames else {}
        hs = vm(pixel_values=px, output_hidden_states=True, **kw).hidden_states[-2]
    want_hf = hs if family == "siglip" else hs[:, 1:]
    spec, w = tower._spec_and_weights()
    want = OV.tower_features(spec, w, px, -2, sel)
    assert rel(want, want_hf) < 1e-5                         # the oracle on the weights read from disk IS the saved HF model
    assert got.shape == want.shape and rel(got, want) < 2e-2 and rel(got, want_hf) < 2e-2      # bf16 engine vs fp32
    t32 = Tower(path, SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature=sel, device=DEV, tower_precision="fp32"))
    assert rel(t32(px), want_hf) < 2e-5                      # the reference-precision engine reproduces HF's fp32 forward


def test_sd_featurizer_built_from_a_diffusers_directory(tmp_path):
    """SDFeaturizer(sd_id = a local diffusers-layout directory) - unet/ vae/ scheduler/ text_encoder/ tokenizer/ read from disk
    (dift_sd.py:226-243) - gives bit-identical features to engines built from the same weights in memory and fed the same token ids."""
    import json
    from test_host_checkpoints import write_diffusers_dir
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM.src.models import dift_sd as DS
    from law_of_vision_representation_in_mllms_amd.sd_engine import SdEngine
    from law_of_vision_representation_in_mllms_amd.text_engine import ClipTextEngine
    spec = SW.tiny_sd_spec()
    # a byte-level CLIP BPE vocabulary (256 byte symbols, their word-final forms, <bos>, <eos>; no merges): a real CLIPTokenizer offline
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    chars = [chr(c) for c in cs]
    vocab = {t: i for i, t in enumerate(chars + [c + "</w>" for c in chars] + ["<|startoftext|>", "<|endoftext|>"])}
    ts = SW.TextSpec(vocab=len(vocab), d=spec.unet.cross_dim, mlp=128, layers=2, heads=1, max_pos=11, act="quick_gelu")   # d = the UNet's cross-attention width
    wu, wv, wt = SW.synthetic_unet(spec.unet, 21, n_up_blocks=len(spec.unet.block_out)), SW.synthetic_vae(spec.vae, 22), SW.synthetic_text(ts, 23)
    root = str(tmp_path / "tiny-sd")
    write_diffusers_dir(root, spec, ts, wu, wv, wt)
    os.makedirs(os.path.join(root, "tokenizer"))
    json.dump(vocab, open(os.path.join(root, "tokenizer", "vocab.json"), "w"))
    open(os.path.join(root, "tokenizer", "merges.txt"), "w").write("#version: 0.2\n")
    json.dump({"model_max_length": 11}, open(os.path.join(root, "tokenizer", "tokenizer_config.json"), "w"))
    feat = DS.SDFeaturizer(root, device=DEV, synthetic=False)
    assert feat.tokenizer is not None and feat.spec.unet == spec.unet and feat.text_spec == ts
    g = torch.Generator().manual_seed(4)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    f = 2 ** (len(spec.vae.block_out) - 1)
    z = spec.vae.latent_channels
    n1, n2 = torch.randn(2, z, 64 // f, 64 // f, generator=g), torch.randn(2, z, 64 // f, 64 // f, generator=g)
    got = feat.forward(img.to(DEV), "a cat", t=1, up_ft_index=0, post_noise=n1.to(DEV), ddim_noise=n2.to(DEV))
    ids = feat.tokenize("a cat")
    assert ids.shape == (1, 11) and int(ids[0, 0]) == vocab["<|startoftext|>"] and int(ids[0, -1]) == vocab["<|endoftext|>"]
    emb = ClipTextEngine(ts, wt, DEV).forward(ids)
    tok = SdEngine(spec, wu, wv, DEV, up_ft_index=0).forward(img.to(DEV), emb, t=1, post_noise=n1.to(DEV), ddim_noise=n2.to(DEV))
    want = tok.view(2, 1, got.shape[-2], got.shape[-1], tok.shape[2]).permute(0, 1, 4, 2, 3).squeeze()
    assert got.shape == want.shape and torch.isfinite(got.float()).all() and torch.equal(got, want)
