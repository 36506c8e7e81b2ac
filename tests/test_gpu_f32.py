"""GPU tests of the reference-precision (fp32) tower path (csrc/f32ops.hip) and of END-TO-END score parity:
images -> tower(fp32) -> mm_projector(fp32) -> A score, and images -> DINOv2 maps -> keypoint transfer -> PCK, each against the fp32
CPU oracle chain at the north-star bar (1e-4 relative on the scores, exact hit counts) - VERDICT r1 row X1.  The reference runs the
C-score CLIP / OpenCLIP / DINOv2 towers in fp32 (C_score/extract_feature.py:36-45,49-50,80-87)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from test_oracle_golden import G, VIT_HIP_TAGS, load_vit_hip_case  # noqa: E402

from law_of_vision_representation_in_mllms_amd import _lib, ascore_ops, cscore_ops, engine  # noqa: E402
from law_of_vision_representation_in_mllms_amd import vit_weights as VW  # noqa: E402
from oracle import ascore as OA, cscore as OC, projector as OP, vit as OV  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (577, 64, 577), (130, 260, 37), (5, 12, 4), (1, 4, 1030), (300, 132, 64)])
def test_gemm_f32_against_float64(M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    ld = (K + 3) // 4 * 4
    a = torch.zeros(M, ld)
    a[:, :K] = torch.randn(M, K, generator=g)
    w = torch.zeros(N, ld)
    w[:, :K] = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    want = (a[:, :K].double() @ w[:, :K].double().t() + bias.double())
    ad, wd = a.to(DEV)[:, :K], w.to(DEV)[:, :K]                              # views with ld % 4 == 0: arbitrary K, aligned rows
    got = engine.gemm_f32(ad, wd, bias.to(DEV))
    assert rel(got, want) < 2e-6
    # [K, N] operand (plain matrix product), N padded to a multiple of 4 in memory
    ldn = (N + 3) // 4 * 4
    wk = torch.zeros(K, ldn)
    wk[:, :N] = w[:, :K].t()
    got = engine.gemm_f32(ad, wk.to(DEV)[:, :N], bias.to(DEV), w_kn=True)
    assert rel(got, want) < 2e-6
    # epilogues: activation kinds, LayerScale + residual (in place), alpha
    for act, ref in (("quick_gelu", lambda x: x * torch.sigmoid(1.702 * x)), ("gelu", torch.nn.functional.gelu),
                     ("gelu_tanh", lambda x: torch.nn.functional.gelu(x, approximate="tanh"))):
        got = engine.gemm_f32(ad, wd, bias.to(DEV), _lib.EPI_ACT, act=act)
        assert rel(got, ref(want)) < 5e-6, act
    res = torch.randn(M, N, generator=g)
    ls = torch.randn(N, generator=g)
    out = res.clone().to(DEV)
    engine.gemm_f32(ad, wd, bias.to(DEV), _lib.EPI_RESID, resid=out, ls=ls.to(DEV), out=out)
    assert rel(out, res.double() + ls.double() * want) < 5e-6
    got = engine.gemm_f32(ad, wd, None, alpha=0.125)
    assert rel(got, 0.125 * (want - bias.double())) < 2e-6


def test_split_bf16x3_carries_24_bits():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(37, 256, generator=g) * torch.logspace(-6, 6, 256)[None]      # 12 decades of magnitudes
    x[3, 5] = 0.0
    p = engine.split_bf16x3(x.to(DEV)).cpu()
    hi, mid, lo = p[:, :256].float(), p[:, 256:512].float(), p[:, 512:].float()
    assert torch.equal(hi, x.to(torch.bfloat16).float())                          # round-to-nearest-even planes
    assert torch.equal(mid, (x - hi).to(torch.bfloat16).float())
    assert torch.equal(lo, (x - hi - mid).to(torch.bfloat16).float())
    err = ((hi.double() + mid.double() + lo.double()) - x.double()).abs()
    assert (err <= x.abs().double() * 2.0 ** -24).all()
    # a strided source (views of a padded buffer) gives the same planes
    buf = torch.zeros(37, 260)
    buf[:, :256] = x
    assert torch.equal(engine.split_bf16x3(buf.to(DEV)[:, :256]).cpu(), p)


def test_split_bf16_two_planes_carry_16_bits():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(19, 128, generator=g) * torch.logspace(-5, 5, 128)[None]
    p2, p3 = engine.split_bf16_planes(x.to(DEV), 2).cpu(), engine.split_bf16_planes(x.to(DEV), 3).cpu()
    assert p2.shape == (19, 256) and torch.equal(p2, p3[:, :256])                 # the first two planes of the triple, nothing else
    err = ((p2[:, :128].double() + p2[:, 128:].double()) - x.double()).abs()
    assert (err <= x.abs().double() * 2.0 ** -17).all()
    with pytest.raises(RuntimeError, match="nplanes"):
        engine.split_bf16_planes(x.to(DEV), 4)


@pytest.mark.parametrize("products", [6, 4, 3])
@pytest.mark.parametrize("M,N,K", [(100, 256, 64), (577 * 3, 768, 768), (1000, 1024, 1024), (300, 256, 4096), (256, 3072, 1024)])
def test_gemm_f32_split_against_float64(M, N, K, products):
    """fp32 GEMM on the bf16 matrix pipe against float64.  Six products (three planes per operand): the rounding noise of an fp32
    accumulation over K, like the exact-fp32 MFMA chain (6e-8 at K = 64 ... 1.3e-6 at K = 4096).  Four / three products (two planes = 16
    significand bits per operand): the planes' own 2^-17 truncation, ~3e-6 whatever K.  fp32 epilogues (bias, exact activations,
    LayerScale + in-place residual) and the plane output (= the split of the fp32 result, bit for bit)."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    want = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    npl = engine.split_planes(products)
    ap, wp = engine.split_bf16_planes(ad, npl), engine.split_bf16_planes(wd, npl)
    gemm_split = lambda *a, **k: engine.gemm_f32_split(*a, products=products, **k)
    got = gemm_split(ap, wp, bd)
    e_split, e_native = rel(got, want), rel(engine.gemm_f32(ad, wd, bd), want)
    print(f"M={M} N={N} K={K} products={products}: split {e_split:.2e}  exact-fp32 MFMA chain {e_native:.2e}")
    if products == 6:
        assert e_split < 2e-6 and e_split < 1.5 * e_native + 5e-8, (e_split, e_native)     # measured: 1.17x the exact chain's error at every K
        tol = 2.0 * e_native + 5e-7                                                        # epilogue cases: the same accumulation noise + libm
    else:
        assert e_split < 1e-5, e_split                                                     # 16-bit operands: 2^-17 per term, random signs
        tol = 2e-5
    for act, ref in (("quick_gelu", lambda x: x * torch.sigmoid(1.702 * x)), ("gelu", torch.nn.functional.gelu),
                     ("gelu_tanh", lambda x: torch.nn.functional.gelu(x, approximate="tanh"))):
        got_a = gemm_split(ap, wp, bd, act=act)
        assert rel(got_a, ref(want)) < tol, act
        assert torch.equal(gemm_split(ap, wp, bd, act=act, planes_out=True), engine.split_bf16_planes(got_a, npl)), act
    res = torch.randn(M, N, generator=g)
    ls = torch.randn(N, generator=g)
    out = res.clone().to(DEV)
    gemm_split(ap, wp, bd, resid=out, ls=ls.to(DEV), out=out)
    assert rel(out, res.double() + ls.double() * want) < tol
    out2 = res.clone().to(DEV)
    gemm_split(ap, wp, None, resid=out2, out=out2)                     # no bias, no LayerScale
    assert rel(out2, res.double() + (want - bias.double())) < tol
    with pytest.raises(RuntimeError, match="N % 256"):
        gemm_split(ap, engine.split_bf16_planes(torch.randn(100, K, device=DEV), npl), bd[:100].contiguous())      # N % 256 != 0
    with pytest.raises(RuntimeError, match="products"):
        engine.gemm_f32_split(ap, wp, bd, products=5)


@pytest.mark.parametrize("products", [6, 4, 3])
@pytest.mark.parametrize("family,image,patch", [("clip", 70, 14), ("dinov2", 154, 14), ("siglip", 48, 16), ("clip", 210, 14), ("dinov2", 266, 14)])
def test_f32_tower_split_route_equals_the_exact_route(family, image, patch, products):
    """The reference-precision tower with its projections and attention as split-bf16 products (the default where shapes allow) against the
    exact-fp32 MFMA route and the fp32 oracle.  Six products: the same bar as the exact route (the two differ by fp32 rounding noise only);
    four / three products (16-bit operands): an order of magnitude above that, three below the bf16 engine."""
    spec = VW.tiny_spec(family, image_size=image, patch=patch, d=256, heads=4, mlp=512, layers=4)           # head width 64, d % 256 == 0
    w = VW.synthetic_weights(spec, seed=11)
    px = torch.from_numpy(np.random.RandomState(3).standard_normal((5, 3, image, image)).astype(np.float32))
    split = engine.VitEngineF32(spec, w, DEV, products=products)
    native = engine.VitEngineF32(spec, w, DEV, gemm="native")
    assert split.gemm == "split" and split.products == products and native.gemm == "native" and native.products is None
    a, b = split.forward(px.to(DEV)), native.forward(px.to(DEV))
    want = OV.tower_features(spec, w, px, select_layer=spec.layers, select_feature="cls_patch")
    es, en = rel(a, want), rel(b, want)
    bar = 5e-6 if products == 6 else 5e-5
    assert es < bar and en < 5e-6 and rel(a, b) < bar, (es, en)
    for n_layers in (0, 1, 3):
        assert rel(split.forward(px.to(DEV), n_layers=n_layers), native.forward(px.to(DEV), n_layers=n_layers)) < bar
    # shapes the 256 x 256 kernel does not take fall back to the exact route; asking for the split route there is an error
    odd = VW.tiny_spec(family, image_size=image, patch=patch, d=128, heads=2, mlp=256, layers=2)
    assert engine.VitEngineF32(odd, VW.synthetic_weights(odd, seed=1), DEV).gemm == "native"
    with pytest.raises(ValueError):
        engine.VitEngineF32(odd, VW.synthetic_weights(odd, seed=1), DEV, gemm="split")


def test_gemm_f32_is_an_exact_fma_chain():
    """v_mfma_f32_32x32x2_f32 accumulates k in order with fused multiply-adds: small-integer operands give the exact integer result
    and the result does not depend on the row / column tile an element falls in."""
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-8, 9, (257, 96), generator=g).float()
    w = torch.randint(-8, 9, (131 * 4, 96), generator=g).float()
    got = engine.gemm_f32(a.to(DEV), w.to(DEV)).cpu()
    assert torch.equal(got, a @ w.t())
    x = torch.randn(300, 64, generator=g)
    full = engine.gemm_f32(x.to(DEV), w[:, :64].contiguous().to(DEV)).cpu()
    part = engine.gemm_f32(x[37:41].contiguous().to(DEV), w[:, :64].contiguous().to(DEV)).cpu()
    assert torch.equal(full[37:41], part)                                     # bit-identical: batch / tile invariance


def test_layernorm_and_softmax_f32():
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(37, 1024, generator=g) * 3 + 5)
    gam, bet = torch.randn(1024, generator=g), torch.randn(1024, generator=g)
    xd, gd, bd = x.to(DEV), gam.to(DEV), bet.to(DEV)                       # named: raw pointers do not keep temporaries alive
    y = torch.empty_like(xd)
    _lib.check(lib.visrep_layernorm_f32(_lib.ptr(xd), 1024, _lib.ptr(gd), _lib.ptr(bd), _lib.ptr(y), 1024, 37, 1024, 1e-5, _lib.stream_ptr()), "ln")
    want = torch.nn.functional.layer_norm(x.double(), (1024,), gam.double(), bet.double(), 1e-5)
    assert rel(y, want) < 1e-6
    s = torch.randn(50, 580, generator=g) * 4
    sd = s.to(DEV)
    _lib.check(lib.visrep_softmax_rows_f32(_lib.ptr(sd), 580, 50, 577, _lib.stream_ptr()), "softmax")
    assert rel(sd[:, :577], torch.softmax(s[:, :577].double(), -1)) < 1e-6
    assert torch.equal(sd[:, 577:].cpu(), s[:, 577:])                         # the padding columns are not touched


# ------------------------------------------------------------------------------------------------ towers vs the reference's goldens
def _tiny_case(tag):
    z = np.load(f"{G}/vit_tiny.npz")
    spec = eval(str(z[f"{tag}.spec"]), {"ViTSpec": VW.ViTSpec})
    flat = {k[len(tag) + 3:]: z[k] for k in z.files if k.startswith(f"{tag}.w.")}
    return spec, VW.unflatten(flat), torch.from_numpy(z[f"{tag}.pixels"]), torch.from_numpy(z[f"{tag}.feat"])


@pytest.mark.parametrize("tag", ["clip_quick", "clip_gelu", "dinov2_native", "dinov2_interp", "siglip"])
def test_f32_tower_matches_the_reference_tower_classes(tag):
    """tests/golden/vit_tiny.npz: outputs of the reference's CLIPVisionTower / DinoV2VisionTower (and HF SiglipVisionModel) themselves,
    head width 32, patch 7 - shapes the bf16 engine cannot take.  The fp32 engine reproduces them to fp32 rounding."""
    spec, w, px, want = _tiny_case(tag)
    if w["pos"].shape[0] != spec.tokens:
        w = dict(w)
        w["pos"] = VW.interpolate_pos(w["pos"], spec.has_cls, spec.grid)
    eng = engine.VitEngineF32(spec, w, DEV)
    hid = eng.forward(px.to(DEV), n_layers=spec.layers - 1)
    feat = hid if spec.family == "siglip" else hid[:, 1:]
    assert feat.dtype == torch.float32 and feat.shape == want.shape
    assert (feat.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), tag


@pytest.mark.parametrize("tag", VIT_HIP_TAGS)
def test_f32_tower_matches_reference_golden_hip_shapes(tag):
    spec, w, px, want = load_vit_hip_case(tag)
    eng = engine.VitEngineF32(spec, w, DEV, products=6)                        # the fp32-rounding bar: the fp32-equivalent product set
    hid = eng.forward(px.to(DEV), n_layers=spec.layers - 1)
    feat = hid if spec.family == "siglip" else hid[:, 1:]
    assert (feat.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), tag
    h3 = engine.VitEngineF32(spec, w, DEV, products=3).forward(px.to(DEV), n_layers=spec.layers - 1)      # 16-bit operands: < 1e-4 (the score bar)
    assert rel(h3 if spec.family == "siglip" else h3[:, 1:], want) < 1e-4, tag
    # and two orders of magnitude closer to the reference than the bf16 engine on the same case
    hb = engine.VitEngine(spec, w, DEV).forward(px.to(DEV), n_layers=spec.layers - 1)
    fb = hb if spec.family == "siglip" else hb[:, 1:]
    assert rel(feat, want) < 0.02 * rel(fb, want)


def test_f32_tower_hidden_states_and_chunk_invariance():
    spec, w, px, _ = load_vit_hip_case("dinov2_interp")
    eng = engine.VitEngineF32(spec, w, DEV, products=6)
    hs = OV.vit_hidden_states(spec, w, px)
    for n in range(spec.layers + 1):
        assert rel(eng.forward(px.to(DEV), n_layers=n), hs[n]) < 5e-6, n
    big = torch.cat([px, px.flip(0), px], 0)
    a = eng.forward(big.to(DEV))
    one = engine.VitEngineF32(spec, w, DEV, max_ws_bytes=1, products=6)        # chunk() == 1: one image per launch sequence
    assert one.chunk() == 1
    b = one.forward(big.to(DEV))
    assert torch.equal(a, b)                                                  # bit-identical whatever the batch split
    assert torch.equal(a[: px.shape[0]], eng.forward(px.to(DEV)))


@pytest.mark.parametrize("family,image,patch", [("clip", 70, 14), ("dinov2", 154, 14), ("siglip", 48, 16)])
def test_fused_f32_attention_matches_the_three_launch_path_and_the_oracle(family, image, patch):
    """Head width 64 runs the fused flash-style fp32 attention (attn_f32_kernel: scores never leave the registers); the three-launch
    path (batched Q K^T -> softmax rows -> P V through HBM, what other head widths run) and the fp32 oracle are the checks.  Token counts
    26 (one partial key tile), 122 (two tiles, the second masked from key 58) and 9."""
    spec = VW.tiny_spec(family, image_size=image, patch=patch, d=128, heads=2, mlp=256, layers=3)           # head width 64
    w = VW.synthetic_weights(spec, seed=5)
    px = torch.from_numpy(np.random.RandomState(3).standard_normal((3, 3, image, image)).astype(np.float32))
    eng = engine.VitEngineF32(spec, w, DEV)
    fused = eng.forward(px.to(DEV))
    old = _lib.load().visrep_debug_f32_attention(1)                          # per-thread diagnostic: the three-launch path
    try:
        unfused = eng.forward(px.to(DEV))
    finally:
        _lib.load().visrep_debug_f32_attention(old)
    want = OV.vit_hidden_states(spec, w, px)[spec.layers]
    assert rel(fused, unfused) < 2e-6 and rel(fused, want) < 5e-6 and rel(unfused, want) < 5e-6
    assert torch.equal(fused, eng.forward(px.to(DEV)))                         # deterministic
    assert torch.equal(fused[1:2], eng.forward(px[1:2].to(DEV)))               # and independent of the batch around an image


def test_tower_class_precision_switch(monkeypatch):
    from law_of_vision_representation_in_mllms_amd.llava.model import llava_arch as LA
    spec = VW.tiny_spec("dinov2", image_size=42, patch=14, d=128, heads=2, mlp=256, layers=3)
    monkeypatch.setattr(VW, "SPECS", {**VW.SPECS, 'facebook/dinov2-large': spec})
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    px = torch.randn(2, 3, 42, 42)
    want = OV.tower_features(spec, VW.synthetic_weights(spec, seed=1), px, -2, "patch")
    for prec, tol, dt in (("fp32", 1e-5, torch.float32), ("bf16", 2e-2, torch.bfloat16)):
        cfg = SimpleNamespace(mm_vision_tower='facebook/dinov2-large', mm_vision_select_layer=-2, mm_vision_select_feature='patch', tower_precision=prec)
        tower = LA.build_function_mapping['facebook/dinov2-large'](cfg)
        assert tower.dtype == dt
        out = tower(px)
        assert out.dtype == px.dtype and rel(out, want) < tol, prec


# ------------------------------------------------------------------------------------------------ end to end: images -> scores
def _stack_weights(spec, seed, hidden, gen):
    w = VW.synthetic_weights(spec, seed=seed)
    p0 = torch.randn(hidden, spec.d, generator=gen) * 0.05
    p2 = torch.randn(hidden, hidden, generator=gen) * 0.05
    b0, b2 = torch.randn(hidden, generator=gen) * 0.02, torch.randn(hidden, generator=gen) * 0.02
    return w, (p0, b0, p2, b2)


@pytest.mark.parametrize("d,route,products", [(128, "native", None), (256, "split", 6), (256, "split", 4), (256, "split", 3)])
def test_end_to_end_a_score_from_images_fp32(d, route, products):
    """images -> tower (hidden_states[-2], CLS dropped) -> mlp2x_gelu projector -> A score, every step fp32 on the device, against the
    same chain on the CPU oracle: 1e-4 relative (the north-star bar); the bf16 engine on the same images is reported beside it.
    Width 128 runs the exact-fp32 MFMA projections, width 256 the split-bf16 ones (the engine's default where the shapes allow)."""
    gen = torch.Generator().manual_seed(11)
    hidden = 256
    kw = dict(d=d, heads=d // 64, mlp=2 * d, layers=3)
    specs = {"clip336": VW.tiny_spec("clip", image_size=56, patch=14, **kw), "clip224": VW.tiny_spec("clip", image_size=42, patch=14, **kw),
             "dino": VW.tiny_spec("dinov2", image_size=42, patch=14, **kw), "siglip": VW.tiny_spec("siglip", image_size=48, patch=16, **kw)}
    n_img = 6
    feats_dev, feats_bf16, feats_cpu = {}, {}, {}
    for i, (name, spec) in enumerate(specs.items()):
        w, (p0, b0, p2, b2) = _stack_weights(spec, 20 + i, hidden, gen)
        px = torch.randn(n_img, 3, spec.image_size, spec.image_size, generator=gen)
        sel = "cls_patch" if spec.family == "siglip" else "patch"
        f_cpu = OV.tower_features(spec, w, px, -2, sel)
        feats_cpu[name] = OP.mlp_gelu(f_cpu, [p0, p2], [b0, b2])
        eng32 = engine.VitEngineF32(spec, w, DEV, products=products)
        assert eng32.gemm == route and eng32.products == products
        hid = eng32.forward(px.to(DEV), n_layers=spec.layers - 1)
        f_dev = hid if spec.family == "siglip" else hid[:, 1:]
        h = engine.gemm_f32(f_dev.reshape(-1, spec.d).contiguous(), p0.to(DEV), b0.to(DEV), _lib.EPI_ACT, act="gelu")
        feats_dev[name] = engine.gemm_f32(h, p2.to(DEV), b2.to(DEV)).view(n_img, -1, hidden)
        assert rel(feats_dev[name], feats_cpu[name]) < (2e-5 if products in (None, 6) else 1e-4), name
        hb = engine.VitEngine(spec, w, DEV).forward(px.to(DEV), n_layers=spec.layers - 1)
        fb = hb if spec.family == "siglip" else hb[:, 1:]
        hb2 = engine.gemm(fb.reshape(-1, spec.d).contiguous(), p0.to(DEV).to(torch.bfloat16), b0.to(DEV), _lib.EPI_ACT, act="gelu")
        feats_bf16[name] = engine.gemm(hb2, p2.to(DEV).to(torch.bfloat16), b2.to(DEV)).view(n_img, -1, hidden)
    for name in ("dino", "siglip", "clip336"):
        want, w336, w224 = OA.a_score(list(feats_cpu[name]), list(feats_cpu["clip336"]), list(feats_cpu["clip224"]))
        s336 = ascore_ops.max_cos_mean(feats_dev[name], feats_dev["clip336"]).double().mean().item()
        s224 = ascore_ops.max_cos_mean(feats_dev[name], feats_dev["clip224"]).double().mean().item()
        got = (s336 + s224) / 2
        assert abs(got - want) <= 1e-4 * abs(want), (name, got, want)
        b336 = ascore_ops.max_cos_mean(feats_bf16[name], feats_bf16["clip336"]).double().mean().item()
        b224 = ascore_ops.max_cos_mean(feats_bf16[name], feats_bf16["clip224"]).double().mean().item()
        # the bf16 engine moves the score by ~1e-3 relative (SURVEY §7 hard part 1): bounded here, reported by tools/precision_report.py
        assert abs((b336 + b224) / 2 - want) <= 2e-2 * abs(want), name


def _pck_chain_cpu(maps, pairs, kps, thr, P):
    """oracle chain on [n, P*P, C] fp32 maps: normalise -> keypoint transfer (window soft-argmax) -> hits at alpha = 0.1 / 0.05 / 0.01."""
    hits = np.zeros(3, np.int64)
    preds = []
    for (i, j), (k1, k2), t in zip(pairs, kps, thr):
        d1 = OC.normalize_feats(maps[i][None])
        d2 = OC.normalize_feats(maps[j][None])
        idx = OC.kpts_to_patch_idx(k1, P)
        xy = OC.keypoint_transfer(d1, d2, idx, P)
        preds.append(xy)
        _, _, h = OC.pair_pck(xy, k1, k2, float(t))                          # pck_train.py:101,149-163
        hits += h.sum(dim=-1).numpy()
    return hits, torch.stack(preds)


def _pck_chain_dev(bank, pairs, kps, thr, P):
    K = kps[0][0].shape[0]
    idx = np.stack([OC.kpts_to_patch_idx(k1, P) for k1, _ in kps]).astype(np.int32)
    i1 = torch.tensor([p[0] for p in pairs])
    i2 = torch.tensor([p[1] for p in pairs])
    nkp = torch.full((len(pairs),), K, dtype=torch.int32)
    xy = cscore_ops.transfer(bank, i1, i2, torch.from_numpy(idx), nkp, P, window=5, layout="pc")
    counts = cscore_ops.pck_counts(xy, torch.stack([k for k, _ in kps]), torch.stack([k for _, k in kps]), torch.tensor(thr, dtype=torch.float64), nkp)
    return counts[:, :3].sum(0).cpu().numpy(), xy.cpu()


def _synthetic_pairs(n_img, n_pairs, K, seed):
    rs = np.random.RandomState(seed)
    pairs = [(int(rs.randint(n_img)), int(rs.randint(n_img))) for _ in range(n_pairs)]
    kps = []
    for _ in range(n_pairs):
        k1 = torch.zeros(K, 3)
        k2 = torch.zeros(K, 3)
        k1[:, :2] = torch.from_numpy(rs.uniform(0, 839, (K, 2)).astype(np.float32))
        k2[:, :2] = torch.from_numpy(rs.uniform(0, 839, (K, 2)).astype(np.float32))
        k1[:, 2] = torch.from_numpy((rs.rand(K) > 0.15).astype(np.float32))
        k2[:, 2] = torch.from_numpy((rs.rand(K) > 0.15).astype(np.float32))
        kps.append((k1, k2))
    thr = rs.uniform(150, 700, n_pairs)
    return pairs, kps, thr


def test_end_to_end_c_score_from_images_fp32_tiny():
    """images -> DINOv2-shaped tower (fp32) -> maps -> transfer + PCK on the device == the oracle chain: predictions to 5e-3 px, EXACT hits."""
    spec = VW.tiny_spec("dinov2", image_size=84, patch=14, d=128, heads=2, mlp=256, layers=3)       # 6 x 6 maps
    w = VW.synthetic_weights(spec, seed=31)
    g = torch.Generator().manual_seed(5)
    px = torch.randn(5, 3, 84, 84, generator=g)
    maps_cpu = OV.tower_features(spec, w, px, -2, "patch")
    maps_dev = engine.VitEngineF32(spec, w, DEV).forward(px.to(DEV), n_layers=spec.layers - 1)[:, 1:].contiguous()
    assert rel(maps_dev, maps_cpu) < 1e-5
    pairs, kps, thr = _synthetic_pairs(5, 12, 9, seed=8)
    want_hits, want_xy = _pck_chain_cpu(maps_cpu, pairs, kps, thr, 6)
    got_hits, got_xy = _pck_chain_dev(maps_dev, pairs, kps, thr, 6)
    assert (got_xy - want_xy).abs().max().item() < 5e-3
    assert np.array_equal(got_hits, want_hits)


@pytest.mark.parametrize("products", [6, 4, 3])
def test_end_to_end_c_score_dinov2_large_full_size_fp32(products):
    """BASELINE configs[3] tower at full size: facebook/dinov2-large geometry (24 layers, d = 1024, LayerScale, position embedding
    interpolated 37 -> 16), hidden_states[-2], on 3 images at 224 px, fp32 on the device vs the fp32 oracle; then the PCK chain on
    the resulting 16 x 16 x 1024 maps: exact hit counts.  The bf16 engine's distance on the same images is asserted loosely beside it."""
    base = VW.SPECS["facebook/dinov2-large"]
    native = base.at_resolution(base.pos_grid * base.patch)
    w0 = VW.synthetic_weights(native, seed=1, n_layers=23)
    spec, w = VW.weights_at_resolution(native, w0, 224)
    rs = np.random.RandomState(4)
    px = torch.from_numpy(rs.standard_normal((3, 3, 224, 224)).astype(np.float32))
    want = OV.tower_features(spec, w, px, select_layer=23, select_feature="patch")
    eng = engine.VitEngineF32(spec, w, DEV, products=products)
    assert eng.gemm == "split" and eng.products == products
    got = eng.forward(px.to(DEV), n_layers=23)[:, 1:].contiguous()
    assert got.shape == (3, 256, 1024)
    e32 = rel(got, want)
    assert e32 < 1e-4, e32
    gb = engine.VitEngine(spec, w, DEV).forward(px.to(DEV), n_layers=23)[:, 1:]
    eb = rel(gb, want)
    assert e32 < 0.02 * eb and eb < 3e-2, (e32, eb)
    pairs, kps, thr = _synthetic_pairs(3, 8, 12, seed=9)
    want_hits, want_xy = _pck_chain_cpu(want, pairs, kps, thr, 16)
    got_hits, got_xy = _pck_chain_dev(got, pairs, kps, thr, 16)
    assert (got_xy - want_xy).abs().max().item() < 5e-3
    assert np.array_equal(got_hits, want_hits)


# ------------------------------------------------------------------------------------------------ PCK where it has hits: corresponding image pairs (round 6)
def structured_pairs(n_pairs, side, K, seed, anno=840):
    """Image pairs that CORRESPOND - the second image is an affine warp (rotation +-8 deg, scale 0.92-1.08, shift +-8 %) of a smooth random
    texture, its key points are the warped source key points - so that the tower's maps really match and PCK has hits to get wrong: the random
    images / random key points of the tests above give 2-3 hits per hundred key points, on which "hit counts are exact" says little."""
    import math
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    imgs, pairs, kps, thr = [], [], [], []
    for n in range(n_pairs):
        lo, mid = torch.randn(1, 3, side // 28, side // 28, generator=g), torch.randn(1, 3, side // 7, side // 7, generator=g)
        a = F.interpolate(lo, size=(side, side), mode="bicubic", align_corners=False) * 1.5 + F.interpolate(mid, size=(side, side), mode="bicubic", align_corners=False) * 0.7
        ang = (torch.rand(1, generator=g).item() - 0.5) * math.radians(16)
        sc = 0.92 + 0.16 * torch.rand(1, generator=g).item()
        tx, ty = [(torch.rand(1, generator=g).item() - 0.5) * 0.16 for _ in range(2)]
        A = torch.tensor([[sc * math.cos(ang), -sc * math.sin(ang), tx], [sc * math.sin(ang), sc * math.cos(ang), ty]])     # target (normalised) -> source
        b = F.grid_sample(a, F.affine_grid(A[None], (1, 3, side, side), align_corners=False), mode="bilinear", padding_mode="reflection", align_corners=False)
        imgs += [a[0], b[0]]
        pairs.append((2 * n, 2 * n + 1))
        k1, k2 = torch.zeros(K, 3), torch.zeros(K, 3)
        src = (torch.rand(K, 2, generator=g) * 0.7 + 0.15) * 2 - 1
        trg = (src - A[:, 2]) @ torch.linalg.inv(A[:, :2]).t()
        ok = (trg.abs() < 0.93).all(dim=1).float()
        k1[:, :2], k2[:, :2] = (src + 1) / 2 * anno, (trg + 1) / 2 * anno
        k1[:, 2] = k2[:, 2] = ok
        kps.append((k1, k2))
        thr.append(300 + 400 * torch.rand(1, generator=g).item())
    return torch.stack(imgs), pairs, kps, np.array(thr)


def test_c_score_on_corresponding_image_pairs_full_size():
    """facebook/dinov2-large geometry at 224 px, 23 layers, on 12 corresponding image pairs x 16 key points: the oracle chain gets ~45 % / 20 % / 2 % of
    the key points within alpha = 0.1 / 0.05 / 0.01 (asserted: the test has hits to lose), dozens of predictions sit within a pixel of a threshold.
    fp32 engine, six products (fp32-equivalent) AND three (the throughput set): predictions within 5e-3 px of the oracle's, hit counts EXACT at all
    three alphas.  The bf16 engine's maps move predictions by pixels; its hit counts are reported and bounded (that is the reference's own
    bf16-vs-fp32 gap, not an error of this path)."""
    base = VW.SPECS["facebook/dinov2-large"]
    spec = base.at_resolution(224)
    os.environ["VISREP_FAST_SYNTHETIC"] = "1"
    try:
        w = VW.synthetic_weights(spec, seed=1, n_layers=23)
    finally:
        os.environ.pop("VISREP_FAST_SYNTHETIC", None)
    px, pairs, kps, thr = structured_pairs(12, 224, 16, seed=3)
    want = OV.tower_features(spec, w, px, select_layer=23, select_feature="patch")
    want_hits, want_xy = _pck_chain_cpu(want, pairs, kps, thr, 16)
    n_vis = int(sum((k1[:, 2] * k2[:, 2]).sum().item() for k1, k2 in kps))
    assert want_hits[0] > 0.3 * n_vis and want_hits[1] > 0.1 * n_vis and want_hits[0] > want_hits[1] > want_hits[2], (want_hits, n_vis)
    for products in (6, 3):
        got = engine.VitEngineF32(spec, w, DEV, products=products).forward(px.to(DEV), n_layers=23)[:, 1:].contiguous()
        got_hits, got_xy = _pck_chain_dev(got, pairs, kps, thr, 16)
        shift = (got_xy - want_xy).abs().max().item()
        print("structured C", (products, rel(got, want), shift, got_hits.tolist(), want_hits.tolist(), n_vis))
        assert shift < 5e-3, (products, shift)
        assert np.array_equal(got_hits, want_hits), (products, got_hits, want_hits)
    gb = engine.VitEngine(spec, w, DEV).forward(px.to(torch.bfloat16).to(DEV), n_layers=23)[:, 1:].float().contiguous()
    b_hits, b_xy = _pck_chain_dev(gb, pairs, kps, thr, 16)
    print("structured C bf16", (rel(gb, want), (b_xy - want_xy).abs().max().item(), b_hits.tolist(), want_hits.tolist()))
    assert np.abs(b_hits - want_hits).max() <= max(3, 0.1 * want_hits[0]), (b_hits, want_hits)
