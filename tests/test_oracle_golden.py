"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from law_of_vision_representation_in_mllms_amd import vit_weights as VW
from oracle import ascore as OA
from oracle import cscore as OC
from oracle import projector as OP
from oracle import vit as OV

G = os.path.join(os.path.dirname(__file__), "golden")


def _case_names(z, suffix):
    return sorted({k.split(".")[0] for k in z.files if k.endswith(suffix)})


# --------------------------------------------------------------------------- A score
@pytest.mark.parametrize("case", ["fp32_small", "fp32_wide", "bf16_inputs"])
def test_ascore_matches_reference(case):
    z = np.load(f"{G}/ascore.npz")
    r336 = [torch.from_numpy(a) for a in z[f"{case}.clip336"]]
    r224 = [torch.from_numpy(a) for a in z[f"{case}.clip224"]]
    n = len(r336)
    # the reference hard-codes 100 tensors; the fixture cycles n distinct images -> weights per image
    reps = np.array([len(range(j, 100, n)) for j in range(n)], dtype=np.float64)
    tol = 1e-6 if "fp32" in case else 1e-2     # bf16 case: reference did its arithmetic in bf16
    for enc in ("clip336", "clip224", "encA", "encB"):
        oth = [torch.from_numpy(a) for a in z[f"{case}.{enc}"]]
        s336 = np.array([OA.max_cos_mean(o, r) for o, r in zip(oth, r336)])
        s224 = np.array([OA.max_cos_mean(o, r) for o, r in zip(oth, r224)])
        got = ((s336 * reps).sum() / 100 + (s224 * reps).sum() / 100) / 2
        want = float(z[f"{case}.result.{enc}"])
        assert abs(got - want) <= tol * max(1.0, abs(want)), (enc, got, want)


@pytest.mark.parametrize("case", ["bf16_inputs", "bf16_wide", "bf16_self"])
def test_ascore_reference_arithmetic_matches_reference_bit_for_bit(case):
    """The reference's IN-DTYPE arithmetic (compute.py on the bf16 tensors it really consumes): the per-op-rounding restatement
    reproduces every per-image value of the reference's own op chain exactly, hence the printed averages; clip336 against itself comes out
    as 1.0078125 in the `bf16_self` case, like the published table's CLIP336 row."""
    z = np.load(f"{G}/ascore.npz")
    n = z[f"{case}.clip336"].shape[0]
    reps = [len(range(j, 100, n)) for j in range(n)]
    per = {}
    for enc in ("clip336", "clip224", "encA", "encB"):
        for ref in ("clip336", "clip224"):
            got = np.array([OA.max_cos_mean_reference_arithmetic(torch.from_numpy(z[f"{case}.{enc}"][j]).to(torch.bfloat16),
                                                                 torch.from_numpy(z[f"{case}.{ref}"][j]).to(torch.bfloat16)) for j in range(n)])
            assert np.array_equal(got, z[f"{case}.per_image.{enc}.{ref}"]), (enc, ref, got, z[f"{case}.per_image.{enc}.{ref}"])
            per[enc, ref] = got
    for enc in ("clip336", "clip224", "encA", "encB"):
        # the script's python-float reduction over its 100 files (the fixture cycles n distinct images), compute.py:75-81
        s336 = [per[enc, "clip336"][i % n] for i in range(100)]
        s224 = [per[enc, "clip224"][i % n] for i in range(100)]
        got = (sum(s336) / 100 + sum(s224) / 100) / 2
        assert abs(got - float(z[f"{case}.result.{enc}"])) <= 1e-12, (enc, got)
    if case == "bf16_self":
        assert np.all(z[f"{case}.per_image.clip336.clip336"] == 1.0078125)


def test_ascore_self_similarity_is_one():
    x = torch.randn(17, 40)
    assert abs(OA.max_cos_mean(x, x) - 1.0) < 1e-6


# --------------------------------------------------------------------------- C score
def test_cscore_transfer_matches_reference():
    z = np.load(f"{G}/cscore_transfer.npz")
    for name in _case_names(z, ".xy"):
        P, C, K, soft, win = z[f"{name}.meta"].tolist()
        f1, f2 = torch.from_numpy(z[f"{name}.f1"]), torch.from_numpy(z[f"{name}.f2"])
        kps = torch.from_numpy(z[f"{name}.kps"])
        idx = OC.kpts_to_patch_idx(kps, P)
        assert np.array_equal(idx, z[f"{name}.patch_idx"]), name
        d1, d2 = OC.descriptors_from_map(f1, P), OC.descriptors_from_map(f2, P)
        xy = OC.keypoint_transfer(d1, d2, idx, P, 840, bool(soft), win).numpy()
        np.testing.assert_allclose(xy, z[f"{name}.xy"], rtol=0, atol=2e-3, err_msg=name)


def test_cscore_pck_matches_reference():
    z = np.load(f"{G}/cscore_pck.npz")
    P, C = z["meta"].tolist()
    per_cat, weights = [], []
    for cat in ("catA", "catB"):
        feats = z[f"{cat}.feats"]
        fi = z[f"{cat}.file_img"]
        kps = torch.from_numpy(z[f"{cat}.kps"])
        thr = z[f"{cat}.thr"].tolist()
        maps = [torch.from_numpy(feats[i]) for i in fi]
        correct, img_correct, preds = OC.category_pck(maps, list(range(len(thr))), kps, thr, P)
        np.testing.assert_allclose(correct, z[f"{cat}.correct"], atol=1e-6)
        np.testing.assert_allclose(img_correct, z[f"{cat}.img_correct"], atol=1e-6)
        K = kps.shape[1]
        np.testing.assert_allclose(torch.stack(preds).numpy(), z[f"{cat}.pred"][:, :K], atol=2e-3)  # renumber_used_points pads to 30
        per_cat.append(img_correct[:3])
        weights.append(img_correct[3])
    w = OC.weighted_pcks(per_cat, weights)
    assert 0 <= w[2] <= w[1] <= w[0] <= 1


# --------------------------------------------------------------------------- ViT towers
@pytest.mark.parametrize("tag", ["clip_quick", "clip_gelu", "dinov2_native", "dinov2_interp", "siglip"])
def test_vit_oracle_matches_hf(tag):
    z = np.load(f"{G}/vit_tiny.npz")
    spec = eval(str(z[f"{tag}.spec"]), {"ViTSpec": VW.ViTSpec})
    flat = {k[len(tag) + 3:]: z[k] for k in z.files if k.startswith(f"{tag}.w.")}
    w = VW.unflatten(flat)
    px = torch.from_numpy(z[f"{tag}.pixels"])
    sel = "cls_patch" if spec.family == "siglip" else "patch"
    feat = OV.tower_features(spec, w, px, select_layer=-2, select_feature=sel)
    want = torch.from_numpy(z[f"{tag}.feat"])
    assert feat.shape == want.shape
    err = (feat - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), (tag, err)


# --------------------------------------------------------------------------- projector
def test_projector_oracle_matches_reference():
    z = np.load(f"{G}/projector.npz")
    x = torch.from_numpy(z["x"])
    y = OP.mlp_gelu(x, [torch.from_numpy(z["w.0.weight"]), torch.from_numpy(z["w.2.weight"])],
                    [torch.from_numpy(z["w.0.bias"]), torch.from_numpy(z["w.2.bias"])])
    np.testing.assert_allclose(y.numpy(), z["y"], atol=1e-5)
    yl = OP.mlp_gelu(x, [torch.from_numpy(z["wl.weight"])], [torch.from_numpy(z["wl.bias"])])
    np.testing.assert_allclose(yl.numpy(), z["y_linear"], atol=1e-5)


def load_vit_hip_case(tag):
    """(spec, weights, pixels, reference features) of a HIP-runnable golden case; weights regenerated from the seed."""
    z = np.load(f"{G}/vit_hip.npz")
    base = eval(str(z[f"{tag}.spec"]), {"ViTSpec": VW.ViTSpec})
    w = VW.synthetic_weights(base, int(z[f"{tag}.seed"]))
    spec, w = VW.weights_at_resolution(base, w, int(z[f"{tag}.res"]))
    return spec, w, torch.from_numpy(z[f"{tag}.pixels"]), torch.from_numpy(z[f"{tag}.feat"])


VIT_HIP_TAGS = ["clip_quick", "clip_gelu", "dinov2_native", "dinov2_interp", "siglip"]


@pytest.mark.parametrize("tag", VIT_HIP_TAGS)
def test_vit_oracle_matches_hf_hip_shapes(tag):
    spec, w, px, want = load_vit_hip_case(tag)
    sel = "cls_patch" if spec.family == "siglip" else "patch"
    feat = OV.tower_features(spec, w, px, select_layer=-2, select_feature=sel)
    assert feat.shape == want.shape
    assert (feat - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


# ------------------------------------------------------------------------------------------------ SD feature tower
SD_TAGS = ("conv_up0", "conv_up1_ens2", "linear_up0", "xl_up0")
_SD_CASES = {"conv_up0": (False, 0, 1, 100, 11), "conv_up1_ens2": (False, 1, 2, 261, 12), "linear_up0": (True, 0, 1, 1, 13),
             "xl_up0": ("xl", 0, 1, 261, 14)}


def load_sd_case(tag):
    """(spec, unet weights, vae weights, inputs dict, expected features) of a tests/golden/sd_tiny.npz case."""
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    linear, idx, ens, t, seed = _SD_CASES[tag]
    z = np.load(os.path.join(G, "sd_tiny.npz"))
    sp = SW.tiny_sdxl_spec() if linear == "xl" else SW.tiny_sd_spec(linear_projection=linear)
    wu, wv = SW.synthetic_unet(sp.unet, seed, n_up_blocks=idx + 1), SW.synthetic_vae(sp.vae, seed + 100)
    inp = {k: torch.from_numpy(z[f"{tag}.{k}"]) for k in ("img", "prompt_embeds", "post_noise", "ddim_noise", "noisy_latents", "mean", "logvar")}
    inp.update(t=t, up_ft_index=idx, ensemble_size=ens)
    return sp, wu, wv, inp, torch.from_numpy(z[f"{tag}.features"])


@pytest.mark.parametrize("tag", SD_TAGS)
def test_sd_oracle_matches_reference(tag):
    from oracle import diffusion as OD
    sp, wu, wv, inp, want = load_sd_case(tag)
    mean, logvar = OD.vae_encode_moments(sp.vae, wv, inp["img"].repeat_interleave(inp["ensemble_size"], dim=0))
    torch.testing.assert_close(mean, inp["mean"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(logvar, inp["logvar"], rtol=1e-4, atol=1e-4)
    noisy = OD.noisy_latents(sp, mean, logvar, inp["post_noise"], inp["ddim_noise"], inp["t"])
    torch.testing.assert_close(noisy, inp["noisy_latents"], rtol=1e-4, atol=1e-5)
    got = OD.sd_features(sp, wu, wv, inp["img"], inp["prompt_embeds"], inp["post_noise"], inp["ddim_noise"], t=inp["t"],
                         up_ft_index=inp["up_ft_index"], ensemble_size=inp["ensemble_size"])
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-3, atol=2e-4)


# ------------------------------------------------------------------------------------------------ CLIP text encoder
TEXT_TAGS = {"quick": ("quick_gelu", 2, 16, 0), "gelu": ("gelu", 3, 77, 1)}


def load_text_case(tag):
    """(spec, weights, input ids, expected last_hidden_state) of a tests/golden/text_tiny.npz case."""
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    act, layers, L, seed = TEXT_TAGS[tag]
    z = np.load(os.path.join(G, "text_tiny.npz"))
    ts = SW.tiny_text_spec(act, layers, L)
    return ts, SW.synthetic_text(ts, seed + 40), torch.from_numpy(z[f"{tag}.ids"]), torch.from_numpy(z[f"{tag}.y"])


@pytest.mark.parametrize("tag", list(TEXT_TAGS))
def test_text_oracle_matches_hf(tag):
    from oracle import text as OT
    ts, w, ids, want = load_text_case(tag)
    got = OT.clip_text_hidden(w, ids, heads=ts.heads, act=ts.act)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
    pen = torch.from_numpy(np.load(os.path.join(G, "text_tiny.npz"))[f"{tag}.penultimate"])
    torch.testing.assert_close(OT.clip_text_hidden(w, ids, heads=ts.heads, act=ts.act, hidden_state=-2), pen, rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------------------------------------ DiT feature tower
DIT_TAGS = {"last": (-1, 261, 31), "first_other_res": (0, 50, 32)}


def load_dit_case(tag):
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    idx, t, seed = DIT_TAGS[tag]
    z = np.load(os.path.join(G, "dit_tiny.npz"))
    sp = SW.tiny_dit_spec()
    inp = {k: torch.from_numpy(z[f"{tag}.{k}"]) for k in ("img", "post_noise", "ddim_noise", "noisy_latents")}
    inp.update(t=t, up_ft_index=idx)
    return sp, SW.synthetic_dit(sp.core, seed), SW.synthetic_vae(sp.vae, seed + 100), inp, torch.from_numpy(z[f"{tag}.features"])


@pytest.mark.parametrize("tag", list(DIT_TAGS))
def test_dit_oracle_matches_reference(tag):
    from oracle import dit as ODT
    sp, wd, wv, inp, want = load_dit_case(tag)
    got = ODT.dit_features(sp, wd, wv, inp["img"], inp["post_noise"], inp["ddim_noise"], t=inp["t"], up_ft_index=inp["up_ft_index"])
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-3, atol=3e-4)


# ------------------------------------------------------------------------------------------------ image-variation tower
def load_imsd_case():
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW, vit_weights as VW
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM.src.models.dift_imsd import synthetic_image_encoder
    z = np.load(os.path.join(G, "imsd_tiny.npz"))
    sp = SW.tiny_sd_spec()
    vs = VW.tiny_spec("clip", image_size=224, patch=56, d=128, layers=2, heads=2, mlp=256)
    enc = synthetic_image_encoder(vs, sp.unet.cross_dim, 51 + 300)
    inp = {k: torch.from_numpy(z[k]) for k in ("img", "post_noise", "ddim_noise", "image_embeds")}
    inp.update(t=261, up_ft_index=0, ensemble_size=2)
    return sp, SW.synthetic_unet(sp.unet, 51, 1), SW.synthetic_vae(sp.vae, 151), vs, enc, inp, torch.from_numpy(z["features"])


def test_imsd_oracle_matches_reference():
    import torch.nn.functional as F
    from oracle import diffusion as OD
    sp, wu, wv, vs, (w, g, b, p), inp, want = load_imsd_case()
    px = F.interpolate(inp["img"], size=(224, 224), mode="bilinear")
    emb = OV.clip_image_embeds(vs, w, g, b, p, px)
    torch.testing.assert_close(emb, inp["image_embeds"], rtol=1e-4, atol=1e-4)
    got = OD.imsd_features(sp, wu, wv, inp["img"], emb.unsqueeze(1), inp["post_noise"], inp["ddim_noise"], t=261, up_ft_index=0, ensemble_size=2)
    torch.testing.assert_close(got, want, rtol=1e-3, atol=3e-4)


# ------------------------------------------------------------------------------------------------ SD3 (MMDiT) feature tower
SD3_TAGS = {"last": (-1, 3, 61), "mid": (1, 1, 62)}


def load_sd3_case(tag):
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    idx, t, seed = SD3_TAGS[tag]
    z = np.load(os.path.join(G, "sd3_tiny.npz"))
    sp = SW.tiny_sd3_spec()
    inp = {k: torch.from_numpy(z[f"{tag}.{k}"]) for k in ("img", "post_noise", "noise", "prompt_embeds", "pooled", "noisy_latents")}
    inp.update(t=t, up_ft_index=idx)
    return sp, SW.synthetic_sd3(sp.core, seed), SW.synthetic_vae(sp.vae, seed + 100), inp, torch.from_numpy(z[f"{tag}.features"])


@pytest.mark.parametrize("tag", list(SD3_TAGS))
def test_sd3_oracle_matches_reference(tag):
    from oracle import diffusion as OD
    from oracle import sd3 as O3
    sp, wc, wv, inp, want = load_sd3_case(tag)
    mean, logvar = OD.vae_encode_moments(sp.vae, wv, inp["img"])
    noisy = O3.flow_noisy_latents(sp, mean, logvar, inp["post_noise"], inp["noise"], inp["t"])
    torch.testing.assert_close(noisy, inp["noisy_latents"], rtol=1e-4, atol=1e-4)
    got = O3.sd3_features(sp, wc, wv, inp["img"], inp["prompt_embeds"], inp["pooled"], inp["post_noise"], inp["noise"], t=inp["t"],
                          up_ft_index=inp["up_ft_index"])
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-3, atol=5e-4)


# ------------------------------------------------------------------------------------------------ supervised post-processor (N4)
def load_aggnet_case(tag):
    import json
    z = np.load(f"{G}/aggnet.npz")
    cfg = json.loads(str(z[f"{tag}.cfg"]))
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}.sd.")}
    return cfg, sd, torch.from_numpy(z[f"{tag}.x"]), torch.from_numpy(z[f"{tag}.y"])


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_aggregation_network_oracle_matches_reference(tag):
    cfg, sd, x, want = load_aggnet_case(tag)
    got = OC.aggregation_network(x, sd, cfg["feature_dims"], cfg["num_norm_groups"])
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # the drop-in module carries the reference's parameter names: a checkpoint of the reference loads without remapping
    from law_of_vision_representation_in_mllms_amd.C_score.model_utils.projection_network import AggregationNetwork
    net = AggregationNetwork(device="cpu", **cfg)
    assert set(net.state_dict()) == set(sd)
    net.load_pretrained_weights(sd)
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k]), k
