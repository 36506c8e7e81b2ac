"""GPU tests of BASELINE.json configs[4] (the 13-setting A + C sweep) and of the multi-process GPU bring-up:
  * a reduced sweep over the ViT-based settings with tiny tower specs, reference precision (fp32 towers + fp32 projector), against the
    CPU oracle chain images -> tower -> projector -> A score / tower -> maps -> PCK: A within 1e-4 relative, PCK to 1e-6;
  * the same sweep in bf16 (the throughput engines), within bf16 tolerances of the oracle;
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


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_reduced_sweep_matches_the_oracle_chain(tiny_registry, precision):
    build = lambda st: S.SettingModel(st, DEV, hidden=HIDDEN, precision=precision, fast_weights=False)
    out = S.run_sweep(SETTINGS, N_A, spair_small(), DEV, build=build, precision=precision)
    assert out["settings"] == 5 and out["images"] == 5 * (N_A + 12) and out["wall_s"] > 0
    want = oracle_sweep(torch.float32 if precision == "fp32" else torch.bfloat16)
    for st in SETTINGS:
        ent, (A, pck) = out["per_setting"][st.name], want[st.name]
        if precision == "fp32":
            assert abs(ent["A"] - A) <= 1e-4 * abs(A), (st.name, ent["A"], A)          # the north-star bar, images -> score
            np.testing.assert_allclose(ent["pck"], pck, atol=1e-6, err_msg=st.name)    # same hits -> same weighted PCK
        else:
            assert abs(ent["A"] - A) <= 2e-2 * abs(A), (st.name, ent["A"], A)
            np.testing.assert_allclose(ent["pck"], pck, atol=0.1, err_msg=st.name)     # a handful of the ~150 key points may flip in bf16


def test_encoder_sharded_a_score_equals_image_sharded_on_device(tiny_registry):
    build = lambda st: S.SettingModel(st, DEV, hidden=HIDDEN, precision="bf16", fast_weights=False)
    a = S.run_sweep(SETTINGS, N_A, None, DEV, build=build, do_c=False)
    b = S.a_scores_encoder_sharded(SETTINGS, N_A, DEV, build=build, chunk=4)
    for st in SETTINGS:
        assert abs(a["per_setting"][st.name]["A"] - b[st.name]) < 1e-12, st.name


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
