"""Tower parity where it had never been stressed: full ViT-L width, 23 layers, on HOSTILE synthetic
weights (vit_weights.hostile_weights: massive-activation channels 50-200x the rest of the residual stream with LayerNorm gains of 0.02 and 3 on
them, LayerScale spread over 1e-5 .. 1, sharp attention heads with x25 logits, one near-constant token at |mean| ~ 25-50 std) - the regime of
real CLIP / DINOv2 checkpoints, none of which exist offline - instead of N(0, 0.02) draws.

What is compared with what:
  * bf16 engine (LayerNorm fold on / off, VISREP_Q_PRESCALE 2 / 1 / 0) against the fp32 oracle, with the oracle's own bf16 run (what the
    reference's model.to(bfloat16) towers do) as the yardstick: overall rel-L2, rel-L2 over the NON-outlier channels (four planted channels hold
    > 95 % of the tensor's energy and would hide everything else) and the worst single token;
  * fp32 engine, split-bf16 x6 / x4 / x3 and the exact route, against the fp32 oracle;
  * images -> hostile towers -> projector -> A score (<= 1e-4 relative) and -> maps -> transfer -> PCK (exact hits): the north-star bars."""
import functools
import os

import numpy as np
import pytest
import torch

from law_of_vision_representation_in_mllms_amd import _lib, ascore_ops, cscore_ops, engine
from law_of_vision_representation_in_mllms_amd import vit_weights as VW
from oracle import ascore as OA, cscore as OC, projector as OP, vit as OV

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = {
    # id: (registry id, input side, layers run (hidden_states[-2]), images)
    "clip_l14_336": ("openai/clip-vit-large-patch14-336", 336, 23, 2),
    "dinov2_l_224": ("facebook/dinov2-large", 224, 23, 2),
    "siglip_b16_224": ("google/siglip-base-patch16-224", 224, 11, 2),
}


@functools.lru_cache(maxsize=None)
def _case(case, profile):
    """(spec, weights, pixels, fp32 oracle hidden state, bf16 oracle hidden state, non-outlier channel index) - oracle runs once per case."""
    name, side, n_layers, n_img = CASES[case]
    spec = VW.SPECS[name].at_resolution(side)              # position table drawn at the run's own grid (the bicubic 37 -> 16 resize of DINOv2's
    os.environ["VISREP_FAST_SYNTHETIC"] = "1"              # table is covered in test_gpu_fullsize.py; here it would smear the planted token row)
    try:
        w = VW.hostile_weights(spec, seed=3, n_layers=n_layers, profile=profile)
    finally:
        os.environ.pop("VISREP_FAST_SYNTHETIC", None)
    px = torch.from_numpy(np.random.RandomState(11).standard_normal((n_img, 3, side, side)).astype(np.float32)).to(torch.bfloat16)
    want = OV.vit_hidden_states(spec, w, px.float(), n_layers=n_layers)[n_layers]
    ref_bf16 = OV.vit_hidden_states(spec, w, px.float(), n_layers=n_layers, dtype=torch.bfloat16)[n_layers].float()
    others = torch.from_numpy(np.setdiff1d(np.arange(spec.d), np.asarray(w["hostile"]["channels"], dtype=np.int64)))
    return spec, w, px, want, ref_bf16, others


def _errs(got, want, others):
    """overall rel-L2, rel-L2 over the non-outlier channels, worst single token over those channels, worst single token after removing each
    token's own mean (what the next LayerNorm sees: the near-constant token's deviation from its mean is all the information it carries)"""
    got, want = got.float().cpu(), want.float().cpu()
    e_all = ((got - want).norm() / want.norm()).item()
    g, t = got[..., others], want[..., others]
    e_oth = ((g - t).norm() / t.norm()).item()
    tok = ((g - t).norm(dim=-1) / t.norm(dim=-1).clamp_min(1e-20))
    gc, tc = g - g.mean(-1, keepdim=True), t - t.mean(-1, keepdim=True)
    ctr = ((gc - tc).norm(dim=-1) / tc.norm(dim=-1).clamp_min(1e-20))
    return e_all, e_oth, tok.max().item(), ctr.max().item()


def test_hostile_weights_are_hostile():
    """the generator does what it says at full width (oracle activations): planted channels >= 30x the median magnitude of the others; the
    near-constant token of a tower without pre-LayerNorm sits at |mean| > 15 std in the hidden state the features are read from"""
    spec, w, px, want, _, others = _case("dinov2_l_224", "outlier")
    ch = torch.tensor(w["hostile"]["channels"])
    assert want[..., ch].abs().median() > 30 * want[..., others].abs().median()
    spec, w, px, want, _, _ = _case("dinov2_l_224", "const")
    tok = w["hostile"]["const_token"]
    assert (want[:, tok].mean(-1).abs() / want[:, tok].std(-1)).min() > 15


@pytest.mark.parametrize("variant", ["default", "no_ln_fold", "q_prescale_1", "q_prescale_0"])
@pytest.mark.parametrize("profile", ["outlier", "const"])
@pytest.mark.parametrize("case", list(CASES))
def test_bf16_tower_on_hostile_weights(case, profile, variant, monkeypatch):
    spec, w, px, want, ref_bf16, others = _case(case, profile)
    n_layers = CASES[case][2]
    if variant.startswith("q_prescale"):
        monkeypatch.setenv("VISREP_Q_PRESCALE", variant[-1])
    eng = engine.VitEngine(spec, w, DEV, fuse_ln=(variant != "no_ln_fold"))
    got = eng.forward(px.to(DEV), n_layers=n_layers)
    assert torch.isfinite(got.float()).all()
    e_all, e_oth, e_tok, e_ctr = _errs(got, want, others)
    r_all, r_oth, r_tok, r_ctr = _errs(ref_bf16, want, others)
    msg = (case, profile, variant, dict(hip=(e_all, e_oth, e_tok, e_ctr), oracle_bf16=(r_all, r_oth, r_tok, r_ctr)))
    print("hostile bf16", msg)
    assert e_all < max(1.5 * r_all, 2e-2), msg
    assert e_oth < max(1.5 * r_oth, 2e-2), msg
    assert e_tok < max(2.0 * r_tok, 5e-2), msg                        # the worst token: twice the reference's own worst token
    assert e_ctr < max(2.0 * r_ctr, 5e-2), msg                        # ... and with every token's mean removed


@pytest.mark.parametrize("products", [6, 4, 3, None])
@pytest.mark.parametrize("profile", ["outlier", "const"])
@pytest.mark.parametrize("case", ["clip_l14_336", "dinov2_l_224"])
def test_f32_tower_on_hostile_weights(case, profile, products):
    """reference-precision engine against the fp32 oracle: the six-product (fp32-equivalent) set and the exact route at fp32 rounding level
    (1e-4 over the non-outlier channels after 23 layers, 1e-3 on the worst token); the two-plane sets (4 / 3 products) are bounded an order
    looser here - whether they may serve the sweep is decided by the score tests below, not by this number."""
    spec, w, px, want, _, others = _case(case, profile)
    n_layers = CASES[case][2]
    eng = engine.VitEngineF32(spec, w, DEV, gemm="native" if products is None else "split", products=products)
    got = eng.forward(px.float().to(DEV), n_layers=n_layers)
    e_all, e_oth, e_tok, e_ctr = _errs(got, want, others)
    print("hostile fp32", (case, profile, products, e_all, e_oth, e_tok, e_ctr))
    tight = products in (6, None)
    assert e_all < (2e-5 if tight else 2e-4), (case, profile, products, e_all)
    assert e_oth < (1e-4 if tight else 1e-3), (case, profile, products, e_oth)
    assert e_tok < (1e-3 if tight else 1e-2), (case, profile, products, e_tok)
    assert e_ctr < (3e-3 if tight else 3e-2), (case, profile, products, e_ctr)      # the near-constant token: |mean| ~ 35 std amplifies by that factor


# ------------------------------------------------------------------------------------------------ images -> scores on hostile towers
def _synthetic_pairs(n_img, n_pairs, K, seed):
    rs = np.random.RandomState(seed)
    pairs = [(int(rs.randint(n_img)), int(rs.randint(n_img))) for _ in range(n_pairs)]
    kps = []
    for _ in range(n_pairs):
        k1, k2 = torch.zeros(K, 3), torch.zeros(K, 3)
        k1[:, :2] = torch.from_numpy(rs.uniform(0, 839, (K, 2)).astype(np.float32))
        k2[:, :2] = torch.from_numpy(rs.uniform(0, 839, (K, 2)).astype(np.float32))
        k1[:, 2] = torch.from_numpy((rs.rand(K) > 0.15).astype(np.float32))
        k2[:, 2] = torch.from_numpy((rs.rand(K) > 0.15).astype(np.float32))
        kps.append((k1, k2))
    return pairs, kps, rs.uniform(150, 700, n_pairs)


def _pck_cpu(maps, pairs, kps, thr, P):
    hits, preds = np.zeros(3, np.int64), []
    for (i, j), (k1, k2), t in zip(pairs, kps, thr):
        d1, d2 = OC.normalize_feats(maps[i][None]), OC.normalize_feats(maps[j][None])
        xy = OC.keypoint_transfer(d1, d2, OC.kpts_to_patch_idx(k1, P), P)
        preds.append(xy)
        hits += OC.pair_pck(xy, k1, k2, float(t))[2].sum(dim=-1).numpy()
    return hits, torch.stack(preds)


def _pck_dev(bank, pairs, kps, thr, P):
    K = kps[0][0].shape[0]
    idx = np.stack([OC.kpts_to_patch_idx(k1, P) for k1, _ in kps]).astype(np.int32)
    i1, i2 = torch.tensor([p[0] for p in pairs]), torch.tensor([p[1] for p in pairs])
    nkp = torch.full((len(pairs),), K, dtype=torch.int32)
    xy = cscore_ops.transfer(bank, i1, i2, torch.from_numpy(idx), nkp, P, window=5, layout="pc")
    counts = cscore_ops.pck_counts(xy, torch.stack([k for k, _ in kps]), torch.stack([k for _, k in kps]), torch.tensor(thr, dtype=torch.float64), nkp)
    return counts[:, :3].sum(0).cpu().numpy(), xy.cpu()


@pytest.mark.parametrize("products", [6, 3])
@pytest.mark.parametrize("profile", ["outlier", "const"])
def test_c_score_from_images_on_hostile_dinov2_large(profile, products):
    """BASELINE configs[3]'s tower at full size on hostile weights: maps 1e-4 (non-outlier channels), predictions 5e-3 px, EXACT PCK hits."""
    spec, w, px, want, _, others = _case("dinov2_l_224", profile)
    want = want[:, 1:].contiguous()
    eng = engine.VitEngineF32(spec, w, DEV, products=products)
    got = eng.forward(px.float().to(DEV), n_layers=23)[:, 1:].contiguous()
    pairs, kps, thr = _synthetic_pairs(px.shape[0], 10, 12, seed=9)
    want_hits, want_xy = _pck_cpu(want, pairs, kps, thr, 16)
    got_hits, got_xy = _pck_dev(got, pairs, kps, thr, 16)
    shift = (got_xy - want_xy).abs().max().item()
    print("hostile C", (profile, products, _errs(got, want, others), shift, got_hits.tolist(), want_hits.tolist()))
    assert shift < 5e-3, (profile, products, shift)
    assert np.array_equal(got_hits, want_hits), (profile, products, got_hits, want_hits)


@pytest.mark.parametrize("products", [6, 3])
def test_a_score_from_images_on_hostile_towers(products):
    """images -> hostile CLIP-L/14-336 / DINOv2-L towers (fp32 engines) -> mlp2x_gelu 1024 -> 4096 -> 4096 -> A score of DINOv2 against the
    CLIP336 stack: <= 1e-4 relative to the oracle chain (the north-star bar); the bf16 engines' number is bounded beside it."""
    gen = torch.Generator().manual_seed(21)
    hidden = 4096
    feats_dev, feats_cpu, feats_bf = {}, {}, {}
    for name in ("clip_l14_336", "dinov2_l_224"):
        spec, w, px, want, _, _ = _case(name, "outlier")
        p0, p2 = torch.randn(hidden, spec.d, generator=gen) * 0.03, torch.randn(hidden, hidden, generator=gen) * 0.015
        b0, b2 = torch.randn(hidden, generator=gen) * 0.02, torch.randn(hidden, generator=gen) * 0.02
        feats_cpu[name] = OP.mlp_gelu(want[:, 1:], [p0, p2], [b0, b2])
        f = engine.VitEngineF32(spec, w, DEV, products=products).forward(px.float().to(DEV), n_layers=23)[:, 1:]
        h = engine.gemm_f32(f.reshape(-1, spec.d).contiguous(), p0.to(DEV), b0.to(DEV), _lib.EPI_ACT, act="gelu")
        feats_dev[name] = engine.gemm_f32(h, p2.to(DEV), b2.to(DEV)).view(px.shape[0], -1, hidden)
        fb = engine.VitEngine(spec, w, DEV).forward(px.to(DEV), n_layers=23)[:, 1:]
        hb = engine.gemm(fb.reshape(-1, spec.d).contiguous(), p0.to(DEV).to(torch.bfloat16), b0.to(DEV), _lib.EPI_ACT, act="gelu")
        feats_bf[name] = engine.gemm(hb, p2.to(DEV).to(torch.bfloat16), b2.to(DEV)).view(px.shape[0], -1, hidden)
    n = feats_cpu["dinov2_l_224"].shape[0]
    want = float(np.mean([OA.max_cos_mean(feats_cpu["dinov2_l_224"][i], feats_cpu["clip_l14_336"][i]) for i in range(n)]))
    got = ascore_ops.max_cos_mean(feats_dev["dinov2_l_224"], feats_dev["clip_l14_336"]).double().mean().item()
    gbf = ascore_ops.max_cos_mean(feats_bf["dinov2_l_224"], feats_bf["clip_l14_336"]).double().mean().item()
    print("hostile A", (products, got, want, abs(got - want) / abs(want), gbf, abs(gbf - want) / abs(want)))
    assert abs(got - want) <= 1e-4 * abs(want), (products, got, want)
    assert abs(gbf - want) <= 2e-2 * abs(want), (gbf, want)


# ------------------------------------------------------------------------------------------------ GroupNorm of the diffusion towers on hostile statistics
@pytest.mark.parametrize("ratio", [10.0, 30.0, 100.0])
@pytest.mark.parametrize("B,HW,C", [(2, 9216, 128), (1, 36864, 128), (2, 2304, 512), (2, 576, 1280)])
def test_groupnorm_with_large_group_means(B, HW, C, ratio):
    """The diffusion towers' GroupNorm (csrc/convnet.hip: per-block partial sums of x and x^2 in fp32, variance = E[x^2] - mean^2) where that
    formula is weakest: every group's |mean| is `ratio` x its standard deviation (VAE activations have such channels; N(0, 1)-like test data
    does not).  Yardstick: torch's fp32 group_norm on the SAME bf16 inputs.  The error of E[x^2] - mean^2 grows like ratio^2 x (fp32 rounding of
    the sums): measured 7e-6 / 1e-4 / 7e-4 relative error of rstd at ratio 10 / 30 / 100 - below the bf16 rounding of the output (1.7e-3 of the
    normalised tensor, the same at every ratio) in all of them; the per-group statistics themselves (mean, rstd) are checked against float64."""
    from law_of_vision_representation_in_mllms_amd import sd_engine as SE
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(int(ratio) + HW + C)
    G = 32
    cpg = C // G
    sign = torch.where(torch.rand(G, generator=g) < 0.5, -1.0, 1.0)
    mean_c = (sign * ratio).repeat_interleave(cpg) * (1 + 0.02 * torch.randn(C, generator=g))        # channels of a group share the large mean
    x = (torch.randn(B, HW, C, generator=g) + mean_c).to(torch.bfloat16)
    gam, bet = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    want = F.silu(F.group_norm(x.float().permute(0, 2, 1), G, gam, bet, 1e-6)).permute(0, 2, 1)
    xd = x.view(B * HW, C).to(DEV)
    got = SE.groupnorm(xd, gam.to(DEV), bet.to(DEV), B, G, 1e-6, True).view(B, HW, C)
    err = ((got.float().cpu() - want).norm() / want.norm()).item()
    st = SE.groupnorm_stats(xd, B, G, 1e-6).cpu().double()                                          # [B, G, 2] = (mean, rstd)
    x64 = x.double().view(B, HW, G, cpg)
    m64 = x64.mean(dim=(1, 3))
    r64 = 1.0 / torch.sqrt(x64.var(dim=(1, 3), unbiased=False) + 1e-6)
    e_mean = ((st[..., 0] - m64).abs() / m64.abs()).max().item()
    e_rstd = ((st[..., 1] - r64).abs() / r64).max().item()
    print("hostile GN", (B, HW, C, ratio, err, e_mean, e_rstd))
    assert e_mean < 1e-5, (B, HW, C, ratio, e_mean)
    assert e_rstd < (3e-4 if ratio <= 30 else 2e-3), (B, HW, C, ratio, e_rstd)     # measured: 7e-6 / 6e-5..1.3e-4 / 1.7e-4..6.7e-4 at ratio 10 / 30 / 100
    assert err < 3e-3, (B, HW, C, ratio, err)                                        # measured 1.5e-3..1.7e-3 everywhere: the bf16 rounding of the OUTPUT


# ------------------------------------------------------------------------------------------------ a diffusion tower on hostile weights
def _hostile_sd_weights(wu, wv, seed=5):
    """The SD family's analogue of hostile_weights: (a) OUTLIER CHANNELS - three output channels of every resnet conv2 (VAE encoder and UNet) scaled
    30-100x with biases of +-(10-30): the residual stream of both networks carries channels 30-100x the others, and the GroupNorm groups they sit
    in have |mean| >> std; (b) GroupNorm gains of 0.02 / 3 on those channels in every norm1 / norm2 behind them; (c) SHARP heads - to_q / to_k of every
    self-attention x4 (logits x16), the VAE mid-block attention's too."""
    rs = np.random.RandomState(seed)
    wu, wv = {k: v.clone() for k, v in wu.items()}, {k: v.clone() for k, v in wv.items()}
    for w in (wu, wv):
        for name in [n for n in w if n.endswith("conv2.weight")]:
            co = w[name].shape[0]
            ch = rs.choice(co, 3, replace=False)
            fac = np.exp(rs.uniform(np.log(30), np.log(100), 3)).astype(np.float32)
            w[name][ch] *= torch.from_numpy(fac)[:, None, None, None]
            w[name.replace("weight", "bias")][ch] = torch.from_numpy((rs.choice([-1.0, 1.0], 3) * rs.uniform(10, 30, 3)).astype(np.float32))
        for name in [n for n in w if n.endswith(("norm1.weight", "norm2.weight")) and "transformer_blocks" not in n]:
            c = w[name].shape[0]
            ch = rs.choice(c, 4, replace=False)
            w[name][ch[:2]] = 0.02
            w[name][ch[2:]] = 3.0
        for name in [n for n in w if n.endswith(("attn1.to_q.weight", "attn1.to_k.weight", "attentions.0.to_q.weight", "attentions.0.to_k.weight"))]:
            w[name] *= 4.0
    return wu, wv


def test_sd_tower_on_hostile_weights():
    """The composed SD featurizer (VAE encoder -> noisy latents -> UNet down / mid / first up blocks) on hostile weights, tiny SD1.5-shaped spec so that
    the fp32 oracle runs in seconds: posterior moments and tower features against the fp32 oracle, bounded by twice the oracle's own bf16 run -
    GroupNorm on groups with huge means, convolutions whose output has 100x outlier channels, attention with x16 logits."""
    from law_of_vision_representation_in_mllms_amd import sd_engine as SE, sd_weights as SW
    from oracle import diffusion as OD
    bf = lambda t: t.to(torch.bfloat16)
    sp = SW.tiny_sd_spec()
    wu, wv = _hostile_sd_weights(SW.synthetic_unet(sp.unet, 21, n_up_blocks=1), SW.synthetic_vae(sp.vae, 22))
    rs = np.random.RandomState(3)
    B, side = 2, 64
    img = torch.from_numpy(rs.uniform(-1, 1, (B, 3, side, side)).astype(np.float32))
    lat = side // 2 ** (len(sp.vae.block_out) - 1)
    Z = sp.vae.latent_channels
    post = torch.from_numpy(rs.standard_normal((B, Z, lat, lat)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((B, Z, lat, lat)).astype(np.float32))
    pe = torch.from_numpy(rs.standard_normal((1, sp.text_len, sp.unet.cross_dim)).astype(np.float32))
    eng = SE.SdEngine(sp, wu, wv, DEV, up_ft_index=0)
    mom, h, w = eng.vae_moments(img.to(DEV))
    untok = lambda y: y.view(B, h, w, -1).permute(0, 3, 1, 2)
    mean, logvar = untok(mom[:, :Z]).float().cpu(), untok(mom[:, Z: 2 * Z]).float().cpu()
    m32, l32 = OD.vae_encode_moments(sp.vae, wv, img)
    m16, l16 = OD.vae_encode_moments(sp.vae, {k: bf(v) for k, v in wv.items()}, bf(img))
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    print("hostile SD moments", (rel(mean, m32), rel(m16, m32), rel(logvar, l32), rel(l16, l32)))
    assert torch.isfinite(mean).all() and torch.isfinite(logvar).all()
    assert rel(mean, m32) < max(2.0 * rel(m16, m32), 2e-2) and rel(logvar, l32) < max(2.0 * rel(l16, l32), 2e-2)
    got = eng.forward(img, pe, t=261, ensemble_size=1, post_noise=post, ddim_noise=ddim)
    want = OD.sd_features(sp, wu, wv, img, pe, post, ddim, t=261)
    ref16 = OD.sd_features(sp, wu, wv, img, pe, post, ddim, t=261, dtype=torch.bfloat16)
    got_tok = got.float().cpu().reshape(want.shape) if got.shape != want.shape else got.float().cpu()
    e_hip, e_ref = rel(got_tok, want), rel(ref16, want)
    print("hostile SD features", (e_hip, e_ref, tuple(got.shape), tuple(want.shape)))
    assert torch.isfinite(got.float()).all()
    assert e_hip < max(2.0 * e_ref, 3e-2), (e_hip, e_ref)


def test_dit_tower_on_hostile_weights():
    """DiT (adaLN-Zero blocks, heads of 72 padded to 128) on hostile weights, tiny spec (the reference-generated fixture's), 4 blocks: three rows of every
    attention output projection and of every MLP fc2 scaled 30-100x with biases of +-(10-30) (outlier channels in the residual stream, which the
    engine's LayerNorm-with-folded-modulation then normalises), the adaLN modulation tables (norm1.linear: shift / scale / gate of both halves) x4 - so that
    (1 + scale) and the gates are far from the identity the engine's folds were only ever exercised near - and to_q / to_k x3 (logits x9)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_oracle_golden import load_dit_case
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    from law_of_vision_representation_in_mllms_amd.dit_engine import DiTEngine
    from oracle import dit as ODT
    sp0, _, _, _, _ = load_dit_case("last")
    from dataclasses import replace
    sp = replace(sp0, core=replace(sp0.core, layers=4))
    wd, wv = SW.synthetic_dit(sp.core, 41, n_layers=4), SW.synthetic_vae(sp.vae, 42)
    rs = np.random.RandomState(6)
    wd = {k: v.clone() for k, v in wd.items()}
    for name in [n for n in wd if n.endswith(("attn1.to_out.0.weight", "ff.net.2.weight"))]:
        ch = rs.choice(wd[name].shape[0], 3, replace=False)
        wd[name][ch] *= torch.from_numpy(np.exp(rs.uniform(np.log(30), np.log(100), 3)).astype(np.float32))[:, None]
        wd[name.replace("weight", "bias")][ch] = torch.from_numpy((rs.choice([-1.0, 1.0], 3) * rs.uniform(10, 30, 3)).astype(np.float32))
    for name in [n for n in wd if n.endswith(("norm1.linear.weight", "norm1.linear.bias"))]:
        wd[name] *= 4.0
    for name in [n for n in wd if n.endswith(("attn1.to_q.weight", "attn1.to_k.weight", "attn1.to_q.bias", "attn1.to_k.bias"))]:
        wd[name] *= 3.0
    B = 2
    side = sp.core.sample_size * 2 ** (len(sp.vae.block_out) - 1)
    img = torch.from_numpy(rs.uniform(-1, 1, (B, 3, side, side)).astype(np.float32))
    lat = sp.core.sample_size
    post = torch.from_numpy(rs.standard_normal((B, 4, lat, lat)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((B, 4, lat, lat)).astype(np.float32))
    got = DiTEngine(sp, wd, wv, DEV, up_ft_index=3).forward(img, t=261, post_noise=post, ddim_noise=ddim)
    want = ODT.dit_features(sp, wd, wv, img, post, ddim, t=261, up_ft_index=3)
    ref16 = ODT.dit_features(sp, wd, wv, img, post, ddim, t=261, up_ft_index=3, dtype=torch.bfloat16)
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    e_hip, e_ref = rel(got, want), rel(ref16, want)
    print("hostile DiT", (e_hip, e_ref, tuple(got.shape), want.abs().max().item()))
    assert got.shape == want.shape and torch.isfinite(got.float()).all()
    assert e_hip < max(2.0 * e_ref, 2e-2), (e_hip, e_ref)


def test_sd3_tower_on_hostile_weights():
    """SD3 (MMDiT: two token streams, joint attention, adaLN-Zero on both) on hostile weights - the reference-generated fixture's tiny spec and inputs,
    weights perturbed: three rows of every output projection / fc2 of BOTH streams x30-100 with biases of +-(10-30), both streams' adaLN modulation tables
    x4, the image stream's to_q / to_k and the context stream's add_q / add_k x3."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_oracle_golden import load_sd3_case
    from law_of_vision_representation_in_mllms_amd.sd3_engine import Sd3Engine
    from oracle import sd3 as O3
    sp, wc, wv, inp, _ = load_sd3_case("last")
    rs = np.random.RandomState(7)
    wc = {k: v.clone() for k, v in wc.items()}
    for name in [n for n in wc if n.endswith(("attn.to_out.0.weight", "attn.to_add_out.weight", "ff.net.2.weight", "ff_context.net.2.weight"))]:
        ch = rs.choice(wc[name].shape[0], 3, replace=False)
        wc[name][ch] *= torch.from_numpy(np.exp(rs.uniform(np.log(30), np.log(100), 3)).astype(np.float32))[:, None]
        wc[name.replace("weight", "bias")][ch] = torch.from_numpy((rs.choice([-1.0, 1.0], 3) * rs.uniform(10, 30, 3)).astype(np.float32))
    for name in [n for n in wc if n.endswith(("norm1.linear.weight", "norm1.linear.bias", "norm1_context.linear.weight", "norm1_context.linear.bias"))]:
        wc[name] *= 4.0
    for name in [n for n in wc if n.split(".")[-2] in ("to_q", "to_k", "add_q_proj", "add_k_proj")]:
        wc[name] *= 3.0
    kw = dict(t=inp["t"], post_noise=inp["post_noise"], ddim_noise=inp["noise"], pooled=inp["pooled"])
    got = Sd3Engine(sp, wc, wv, DEV, up_ft_index=inp["up_ft_index"]).forward(inp["img"], inp["prompt_embeds"], **kw)
    oa = (sp, wc, wv, inp["img"], inp["prompt_embeds"], inp["pooled"], inp["post_noise"], inp["noise"])
    want = O3.sd3_features(*oa, t=inp["t"], up_ft_index=inp["up_ft_index"])
    ref16 = O3.sd3_features(*oa, t=inp["t"], up_ft_index=inp["up_ft_index"], dtype=torch.bfloat16)
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    e_hip, e_ref = rel(got, want), rel(ref16, want)
    print("hostile SD3", (e_hip, e_ref, tuple(got.shape), want.abs().max().item()))
    assert got.shape == want.shape and torch.isfinite(got.float()).all()
    assert e_hip < max(2.0 * e_ref, 2e-2), (e_hip, e_ref)
