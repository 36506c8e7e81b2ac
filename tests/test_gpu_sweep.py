"""GPU tests of BASELINE.json configs[4] (the 13-setting A + C sweep) and of the multi-process GPU bring-up:
  * a reduced sweep over the ViT-based settings with tiny tower specs in the three precision modes - 'reference' (per leg what the
    reference runs: A leg bf16, C leg fp32 for CLIP / OpenCLIP / DINOv2), 'fp32' (fp32 towers + fp32 projector on both legs) and 'bf16' -
    against the CPU oracle chain images -> tower -> projector -> A score / tower -> maps -> PCK: fp32 legs A within 1e-4 relative and PCK to
    1e-6, bf16 legs within bf16 tolerances;
  * when >= 2 GPUs are visible: bench.py's protocol and the sweep on 2 ranks over RCCL (skipped on 1-GPU boxes), so that the first
    multi-GPU lease is not also the first RCCL bring-up."""
import json
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))


from law_of_vision_representation_in_mllms_amd import sweep as S  # noqa: E402
from law_of_vision_representation_in_mllms_amd import vit_weights as VW  # noqa: E402
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_projector.builder import build_vision_projector  # noqa: E402
from oracle import ascore as OA, cscore as OC, projector as OP, vit as OV  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIDDEN = 256

TINY = {S.CLIP336: VW.tiny_spec("clip", image_size=56, patch=14, d=128, heads=2, mlp=256, layers=3),
        S.CLIP224: VW.tiny_spec("clip", image_size=42, patch=14, d=128, heads=2, mlp=256, layers=3),
        S.DINOV2: VW.tiny_spec("dinov2", image_size=42, patch=14, d=128, heads=2, mlp=256, layers=3),
        S.SIGLIP: VW.tiny_spec("siglip", image_size=48, patch=16, d=128, heads=2, mlp=256, layers=3)}
SETTINGS = (S.Setting("CLIP336", "clip336", (S.CLIP336,), 56, 4), S.Setting("CLIP224", "clip224", (S.CLIP224,), 42, 5),
            S.Setting("DINOv2", "dino", (S.DINOV2,), 42, 3), S.Setting("SigLIP", "siglip", (S.SIGLIP,), 48, 4),
            S.Setting("CLIP224+DINOv2", "clip224+dino", (S.CLIP224, S.DINOV2), 42, 4))
N_A = 9


def spair_small():
    return S.synthetic_spair(n_images=12, n_pairs=30, kmax=8, seed=4, categories=("bird", "cow"))


def oracle_tokens(st, px):
    f = []
    for tid in st.towers:
        spec = TINY[tid]
        f.append(OV.tower_features(spec, VW.synthetic_weights(spec, seed=1), px.float().cpu(), -2, "cls_patch" if spec.family == "siglip" else "patch"))
    return torch.cat(f, -1)


def oracle_projector(width):
    torch.manual_seed(7)                                                        # SettingModel's seed: the same nn.Linear initialisation
    p = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=width, hidden_size=HIDDEN))
    return lambda x: OP.mlp_gelu(x, [p[0].weight.detach(), p[2].weight.detach()], [p[0].bias.detach(), p[2].bias.detach()])


def oracle_sweep(dtype):
    pix = lambda ids, size: S.synthetic_pixels(ids, size, DEV, dtype)
    feats, out = {}, {}
    for st in SETTINGS:
        proj = oracle_projector(128 * len(st.towers))
        feats[st.key] = proj(oracle_tokens(st, pix(range(N_A), st.size)))
    for st in SETTINGS:
        A = OA.a_score(list(feats[st.key]), list(feats["clip336"]), list(feats["clip224"]))[0]
        per_cat, weights = [], []
        for ci, cat in enumerate(spair_small()):
            maps = oracle_tokens(st, pix([ci * 100000 + i for i in range(cat.n_images)], st.size))
            P = int(round(maps.shape[1] ** 0.5))
            fl = []
            for s in cat.slot:
                m = maps[int(s)]
                if len(st.towers) == 2:
                    m = OC.normalize_feats_two(m[None], 128)[0]
                fl.append(m.t().reshape(1, -1, P, P))
            N = len(cat.thresholds)
            per_cat.append(OC.category_pck(fl, list(range(N)), cat.kps, cat.thresholds, P)[1][:3])
            weights.append(N)
        out[st.name] = (A, OC.weighted_pcks(per_cat, weights))
    return out


@pytest.fixture
def tiny_registry(monkeypatch):
    monkeypatch.setattr(VW, "SPECS", {**VW.SPECS, **TINY})


@pytest.mark.parametrize("precision", ["reference", "fp32", "bf16"])
def test_reduced_sweep_matches_the_oracle_chain(tiny_registry, precision):
    """'reference' = the reference's own arithmetic per leg (A leg bf16 like LLaVA, C leg fp32 for CLIP / DINOv2 and the fusion, bf16 for
    SigLIP): its fp32 C legs must give the fp32 oracle chain's PCK exactly, its A scores sit at the bf16 engines' distance."""
    build = lambda st: S.SettingModel(st, DEV, hidden=HIDDEN, precision=precision, fast_weights=False)
    out = S.run_sweep(SETTINGS, N_A, spair_small(), DEV, build=build, precision=precision, also_bf16=True)
    assert out["settings"] == 5 and out["images"] == 5 * (N_A + 12) and out["wall_s"] > 0 and out["tower_precision"] == precision
    want32 = oracle_sweep(torch.float32)
    want = want32 if precision != "bf16" else oracle_sweep(torch.bfloat16)
    for st in SETTINGS:
        ent, (A, pck) = out["per_setting"][st.name], want[st.name]
        c_fp32 = precision == "fp32" or (precision == "reference" and st.name != "SigLIP")
        # fp32 legs are labelled with their route: "fp32" (exact-fp32 MFMA) or "fp32[split-bf16 xN]" (N plane-pair products)
        kind = lambda lab: "+".join(sorted({x.split("[")[0] for x in lab.split("+")}))
        assert {k: kind(v) for k, v in ent["dtype"].items()} == {"a": "fp32" if precision == "fp32" else "bf16", "c": "fp32" if c_fp32 else "bf16"}, st.name
        if precision == "fp32":
            assert abs(ent["A"] - A) <= 1e-4 * abs(A), (st.name, ent["A"], A)          # the north-star bar, images -> score
        else:
            assert abs(ent["A"] - A) <= 2e-2 * abs(A), (st.name, ent["A"], A)
        if c_fp32:
            np.testing.assert_allclose(ent["pck"], pck, atol=1e-6, err_msg=st.name)    # same hits -> same weighted PCK
        else:
            np.testing.assert_allclose(ent["pck"], pck, atol=0.1, err_msg=st.name)     # a handful of the ~150 key points may flip in bf16
        if precision == "reference":
            assert ("c_s_bf16" in ent) == (st.name != "SigLIP")                        # the bf16 twin of every fp32 C leg, timed beside it
    assert ("wall_s_all_bf16" in out) == (precision == "reference")


@pytest.mark.parametrize("world", [4, 3])
def test_c_leg_exchange_with_device_buffers_at_world_4(monkeypatch, world):
    """the C leg's owner-addressed all_to_all_single with REAL device buffers at world > 2.  RCCL refuses two ranks on one
    GPU and gloo has no device all-to-all, so the ranks run as threads of this process (tests/_thread_dist.ThreadDist: rendezvous + device
    copies with the production split tables); everything else is sweep.c_score_of as the launcher runs it - shard plan, launch plan, send /
    receive splits, bank rows, asynchronous completion under the next launch, rank-owned categories, all_gather_object of the results.
    The tower is an ELEMENTWISE function of the pixels (bit-identical whatever the launch's batch), so every rank's result must equal the
    single-process run exactly, and every map crosses the fabric at most once."""
    from _thread_dist import ThreadDist
    from law_of_vision_representation_in_mllms_amd.C_score import pck_train as PT
    P, C_ = 6, 64
    spair = S.synthetic_spair(46, 120)                                   # 18 categories, uneven image counts

    class Tower:
        setting = S.Setting("Elementwise", "ew", ("ew",), 12, 5)        # launches of 5: several exchanges per rank, a remainder launch
        split = 0

        @staticmethod
        def tokens(px):                                                  # [B, 3, 12, 12] -> [B, 36, 64] bf16: sin of (pooled pixel x frequency)
            m = px.float().reshape(px.shape[0], 3, P, 2, P, 2).mean((1, 3, 5)).reshape(px.shape[0], P * P, 1)
            f = torch.arange(1, C_ + 1, device=px.device, dtype=torch.float32).view(1, 1, C_)
            return torch.sin(m * f * 3.0 + f).to(torch.bfloat16)
    pixels = lambda ids, size: S.synthetic_pixels(ids, size, DEV, torch.float32)
    want = S.c_score_of(Tower, spair, pixels, torch.device(DEV), 0, 1)
    td = ThreadDist(world)
    monkeypatch.setattr(S, "_dist", lambda: td)
    monkeypatch.setattr(PT, "_dist", lambda: td)
    got = td.run(lambda r: S.c_score_of(Tower, spair, pixels, torch.device(DEV), r, world))
    for r in range(world):
        assert list(got[r]) == list(want), (r, got[r], want)            # every rank: the single-process numbers, exactly
    n_items = sum(c.n_images for c in spair)
    assert 0 < td.bytes_on_fabric <= n_items * P * P * C_ * 2           # bf16 rows, each at most once


def test_encoder_sharded_a_score_equals_image_sharded_on_device(tiny_registry):
    build = lambda st: S.SettingModel(st, DEV, hidden=HIDDEN, precision="bf16", fast_weights=False)
    a = S.run_sweep(SETTINGS, N_A, None, DEV, build=build, do_c=False)
    b = S.a_scores_encoder_sharded(SETTINGS, N_A, DEV, build=build, chunk=4)
    for st in SETTINGS:
        assert abs(a["per_setting"][st.name]["A"] - b[st.name]) < 1e-12, st.name


# ------------------------------------------------------------------------------------------------ all thirteen setting KINDS
def _tiny_diffusion_registry(monkeypatch):
    """Tiny architectures behind the six diffusion tower ids (synthetic weights), and a noise source that depends on the image only."""
    from dataclasses import replace
    from law_of_vision_representation_in_mllms_amd import sd_engine as SE, sd_weights as SW
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM import diffusion_encoder as DE
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM.src.models import dift_sd as DS, dift_sd3 as D3
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    txt = lambda act="quick_gelu": SW.TextSpec(vocab=99, d=64, mlp=128, layers=2, heads=1, max_pos=11, act=act)
    xl = SW.tiny_sdxl_spec()
    xl = replace(xl, unet=replace(xl.unet, cross_dim=128))                     # two 64-wide text encoders, concatenated
    monkeypatch.setitem(SW.SD_SPECS, S.SD15, SW.tiny_sd_spec("tiny-sd15"))
    monkeypatch.setitem(SW.SD_SPECS, S.SD21, SW.tiny_sd_spec("tiny-sd21", linear_projection=True))
    monkeypatch.setitem(SW.SD_SPECS, S.SDXL, xl)
    for k, v in ((S.SD15, txt()), (S.SD21, txt("gelu")), (S.SDXL, txt())):
        monkeypatch.setitem(DS._SYNTH_TEXT, k, v)
    monkeypatch.setitem(DS._SYNTH_TEXT_2, S.SDXL, txt("gelu"))
    monkeypatch.setitem(SW.DIT_SPECS, S.DIT, SW.tiny_dit_spec())
    s3 = SW.tiny_sd3_spec()
    monkeypatch.setitem(SW.SD3_SPECS, S.SD3, replace(s3, core=replace(s3.core, joint_dim=128, pooled_dim=128)))
    monkeypatch.setattr(D3, "_SYNTH_TEXT", (txt(), txt("gelu")))
    widths = {S.SD15: 128, S.SD21: 128, S.SDXL: 128, S.IMSD: 128, S.DIT: 4 * 8 * 72, S.SD3: 4 * 2 * 64}
    for k, v in widths.items():
        monkeypatch.setitem(DE.feature_hid_size_mapping, k, v)

    def noise(x, shape):                                                        # per image: a fixed function of its pixels
        out = []
        for k in range(2):
            rows = []
            for i in range(shape[0]):
                g = torch.Generator(device=x.device).manual_seed(int(x[i].float().abs().sum().item() * 64) % (2 ** 31) + k)
                rows.append(torch.randn(shape[1:], generator=g, device=x.device))
            out.append(torch.stack(rows))
        return out
    monkeypatch.setattr(SE, "NOISE_FN", noise)


ALL13 = SETTINGS[:2] + (S.Setting("OpenCLIP", "openclip", (S.OPENCLIP,), 42, 4), SETTINGS[2], S.Setting("SDim", "imsd", (S.IMSD,), 64, 3),
                        S.Setting("SD1.5", "sd1.5", (S.SD15,), 64, 4), S.Setting("SDXL", "sdxl", (S.SDXL,), 64, 3), S.Setting("DiT", "dit", (S.DIT,), 64, 4),
                        S.Setting("SD3", "sd3", (S.SD3,), 64, 3), S.Setting("SD2.1", "sd2.1", (S.SD21,), 64, 4), SETTINGS[3], SETTINGS[4],
                        S.Setting("CLIP336+DINOv2", "clip336+dino", (S.CLIP336, S.DINOV2), 56, 3))


@pytest.mark.parametrize("precision", ["reference", "bf16"])
def test_sweep_covers_all_thirteen_setting_kinds(tiny_registry, monkeypatch, precision):
    """the sweep test covered 5 of the 13 settings.  All thirteen kinds (policy/fit.py:20) with tiny architectures: every
    setting's A score and weighted PCK out of run_sweep (one batched, image-sharded tower pass, banks scattered by category owner) must
    equal the SAME towers run one image at a time through the oracle score functions - which pins the sweep's own plumbing (launch plan
    and its tails, '.'-fusion concat and `split`, bank rows, map layout) for the diffusion and 336-fusion settings too."""
    monkeypatch.setitem(VW.SPECS, S.OPENCLIP, VW.tiny_spec("clip", act="gelu", image_size=42, patch=14, d=128, heads=2, mlp=256, layers=3))
    monkeypatch.setitem(TINY, S.OPENCLIP, VW.SPECS[S.OPENCLIP])
    _tiny_diffusion_registry(monkeypatch)
    assert [s.name for s in ALL13] == [s.name for s in S.SETTINGS]
    models = {}

    def build(st):
        models[st.name] = S.SettingModel(st, DEV, hidden=HIDDEN, precision=precision, fast_weights=False)
        return models[st.name]
    spair = spair_small()
    out = S.run_sweep(ALL13, N_A, spair, DEV, build=build, precision=precision)
    assert out["settings"] == 13
    fp32_c = {"CLIP336", "CLIP224", "OpenCLIP", "DINOv2", "CLIP224+DINOv2", "CLIP336+DINOv2"} if precision == "reference" else set()
    assert {n for n, e in out["per_setting"].items() if e["dtype"]["c"] == "fp32"} == fp32_c       # extract_feature.py:36-50,80-91
    assert all(e["dtype"]["a"] == "bf16" for e in out["per_setting"].values())
    pix = lambda ids, size: S.synthetic_pixels(ids, size, DEV, torch.bfloat16 if precision == "bf16" else torch.float32)
    one = lambda m, gid: m.tokens(pix([gid], m.setting.size))                   # batch of one: no launch plan, no bank
    one_c = lambda m, gid: m.c_tokens(pix([gid], m.setting.size))               # ... on the C leg's engines
    feats = {st.key: torch.cat([models[st.name].project(one(models[st.name], i)).float().cpu() for i in range(N_A)]) for st in ALL13}
    for st in ALL13:
        m, ent = models[st.name], out["per_setting"][st.name]
        A = OA.a_score(list(feats[st.key]), list(feats["clip336"]), list(feats["clip224"]))[0]
        assert abs(ent["A"] - A) <= 2e-3 * abs(A) + 1e-4, (st.name, ent["A"], A)   # bf16 features; the sweep's score kernel vs the fp64 oracle
        per_cat, weights = [], []
        for ci, cat in enumerate(spair):
            maps = torch.cat([one_c(m, ci * 100000 + i).float().cpu() for i in range(cat.n_images)])
            P = int(round(maps.shape[1] ** 0.5))
            fl = []
            for sl in cat.slot:
                mp = maps[int(sl)]
                if len(st.towers) == 2:
                    mp = OC.normalize_feats_two(mp[None], m.split)[0]
                fl.append(mp.t().reshape(1, -1, P, P))
            per_cat.append(OC.category_pck(fl, list(range(len(cat.thresholds))), cat.kps, cat.thresholds, P)[1][:3])
            weights.append(len(cat.thresholds))
        # bf16 legs: <= 1-2 of ~150 key points may flip between a batched and a single-image launch; fp32 legs are batch-invariant bit for bit
        np.testing.assert_allclose(ent["pck"], OC.weighted_pcks(per_cat, weights), atol=1e-6 if st.name in fp32_c else 0.04, err_msg=st.name)


def _torchrun(n, *cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 300
    full = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port), *cmd]
    r = subprocess.run(full, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL bring-up)")
def test_two_rank_bench_protocol_over_rccl():
    line = _torchrun(2, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32", "--no-cpu-baseline", "--sweep", "off")
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["value"] > 0 and line["scaling"] == "weak"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL bring-up)")
def test_two_rank_sweep_over_rccl_equals_one_rank():
    args = ["-m", "law_of_vision_representation_in_mllms_amd.sweep", "--a-images", "9", "--c-images", "36", "--c-pairs", "72", "--settings", "CLIP336", "CLIP224",
            "SigLIP", "CLIP224+DINOv2"]
    two = _torchrun(2, *args)
    one = _torchrun(1, *args)
    assert two["world"] == 2 and one["world"] == 1
    for name, ent in one["per_setting"].items():
        assert abs(two["per_setting"][name]["A"] - ent["A"]) <= 1e-9 * abs(ent["A"]), name
        np.testing.assert_allclose(two["per_setting"][name]["pck"], ent["pck"], atol=1e-12, err_msg=name)
