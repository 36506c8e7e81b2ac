"""AP-10k and PF-Pascal evaluation sets on CPU: loaders, category/split selection, eval() numbers and log lines
against the reference's own run on synthetic mini trees (tests/golden/mini_ap10k + nextsets.npz, made by make_golden.py
nextsets).  Device entry points are replaced by the oracle-backed stand-ins of test_host_cscore."""
import argparse
import logging
import os
import shutil

import numpy as np
import pytest
import torch
from PIL import Image

from law_of_vision_representation_in_mllms_amd.C_score import pck_train as PT
from law_of_vision_representation_in_mllms_amd.C_score.utils import utils_dataset as UD
from test_host_cscore import cpu_ops  # noqa: F401 (fixture)

G = os.path.join(os.path.dirname(__file__), "golden")
SUBSETS = ("intra-species", "cross-species", "cross-family")


def make_trees(tmp):
    """<tmp>/data/ap-10k and <tmp>/data/PF-dataset-PASCAL with feature files; returns the fixture archive."""
    z = np.load(f"{G}/nextsets.npz")
    shutil.copytree(f"{G}/mini_ap10k", f"{tmp}/data/ap-10k")
    pr = f"{tmp}/data/PF-dataset-PASCAL"
    os.makedirs(f"{pr}/JPEGImages")
    os.makedirs(f"{pr}/features")
    for key in z.files:
        if key.startswith("ap10k.feat."):
            _, _, fam, species, name = key.split(".")
            os.makedirs(f"{tmp}/data/ap-10k/features/{fam}/{species}", exist_ok=True)
            torch.save(torch.from_numpy(z[key]), f"{tmp}/data/ap-10k/features/{fam}/{species}/{name}_dino.pt")
        if key.startswith("pascal.feat."):
            torch.save(torch.from_numpy(z[key]), f"{pr}/features/{key.split('.')[2]}_dino.pt")
    for c in z["pascal.cats"]:
        os.makedirs(f"{pr}/Annotations/{c}")
    with open(f"{pr}/test_pairs_pf_pascal.csv", "w") as f:
        f.write(str(z["pascal.csv"]))
    for name, (w, h) in zip(z["pascal.images"], z["pascal.sizes"]):
        Image.new("RGB", (int(w), int(h))).save(f"{tmp}/data/{name}")
    return z


def base_args(**kw):
    a = dict(NUM_PATCHES=16, COMPUTE_GEOAWARE_METRICS=True, ADAPT_FLIP=False, EVAL_DATASET="ap10k", TRAIN_DATASET="spair", ANNO_SIZE=840,
             ENSEMBLE=1, MODEL="dino", SOFT_EVAL=True, SOFT_EVAL_WINDOW=5, KPT_RESULT=True, TOTAL_SAVE_RESULT=0, MUTUAL_NN=False,
             TEST_SAMPLE=0, BBOX_THRE=True, AP10K_EVAL_SUBSET="intra-species")
    a.update(kw)
    return argparse.Namespace(**a)


@pytest.mark.parametrize("subset", SUBSETS)
def test_ap10k_loader_and_categories_match_reference(tmp_path, monkeypatch, subset):
    z = make_trees(str(tmp_path))
    monkeypatch.chdir(tmp_path)                                   # the pair files hold cwd-relative annotation paths
    d, cats, split = UD.get_dataset_info(base_args(AP10K_EVAL_SUBSET=subset), "test")
    assert (d, cats, split) == ("data/ap-10k", list(z[f"ap10k.{subset}.cats"]), str(z[f"ap10k.{subset}.split"]))
    for c in cats:
        files, kps, thr, used = UD.load_ap10k_data(d, 840, c, split, 0)
        assert files == list(z[f"ap10k.{subset}.{c}.files"])
        assert torch.equal(kps, torch.from_numpy(z[f"ap10k.{subset}.{c}.kps"]))
        np.testing.assert_array_equal(np.asarray(thr, np.float64), z[f"ap10k.{subset}.{c}.thr"])
        np.testing.assert_array_equal(used.numpy(), z[f"ap10k.{subset}.{c}.used"])
    if subset == "cross-family":                                  # TEST_SAMPLE > 0: seeded draw with replacement
        assert UD.load_ap10k_data(d, 840, "all", split, 6)[0] == list(z["ap10k.sub.files"])


@pytest.mark.parametrize("subset", SUBSETS)
def test_ap10k_eval_matches_reference_eval(tmp_path, monkeypatch, cpu_ops, caplog, subset):
    z = make_trees(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    scores = []
    spy = lambda *a, **k: (lambda r: (scores.append(r[1]), r)[1])(PT.compute_pck(*a, **k))
    with caplog.at_level(logging.INFO, logger="visrep.cscore"):
        r = PT.eval(base_args(AP10K_EVAL_SUBSET=subset), PT.DummyAggregationNetwork(), str(tmp_path), split="test", _compute=spy)
    np.testing.assert_allclose(r[:3], z[f"ap10k.{subset}.pck"], atol=1e-7)
    np.testing.assert_allclose(np.array(scores, np.float64), z[f"ap10k.{subset}.scores"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.stack([x["src_kpts_pred"] for x in r[3]]), z[f"ap10k.{subset}.pred"], atol=2e-3)
    assert caplog.messages == list(z[f"ap10k.{subset}.log"])


def test_pascal_loader_matches_reference(tmp_path, monkeypatch):
    z = make_trees(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    a = base_args(EVAL_DATASET="pascal")
    d, cats, split = UD.get_dataset_info(a, "test")
    assert (d, cats, split) == ("data/PF-dataset-PASCAL", list(z["pascal.cats"]), "test")
    for c in cats:
        files, kps, thr, used = UD.load_eval_data(a, d, c, split)
        assert thr is None and files == list(z[f"pascal.{c}.files"])
        assert torch.equal(kps, torch.from_numpy(z[f"pascal.{c}.kps"]))
        np.testing.assert_array_equal(used.numpy(), z[f"pascal.{c}.used"])
    with pytest.raises(NotImplementedError):
        UD.load_pascal_data(d, 840, "cat", "train", 0)


@pytest.mark.parametrize("tag,kpt", [("img", False), ("kpt", True)])
def test_pascal_eval_matches_reference_eval(tmp_path, monkeypatch, cpu_ops, caplog, tag, kpt):
    """No bbox thresholds: hits are err < float32(alpha * ANNO_SIZE) with alphas (0.1, 0.05, 0.15)."""
    z = make_trees(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    a = base_args(EVAL_DATASET="pascal", COMPUTE_GEOAWARE_METRICS=False, BBOX_THRE=False, KPT_RESULT=kpt)
    with caplog.at_level(logging.INFO, logger="visrep.cscore"):
        r = PT.eval(a, PT.DummyAggregationNetwork(), str(tmp_path), split="test")
    np.testing.assert_allclose(r[:3], z[f"pascal.{tag}.pck"], atol=1e-7)
    np.testing.assert_allclose(np.stack([x["src_kpts_pred"] for x in r[3]]), z["pascal.pred"], atol=2e-3)
    assert caplog.messages == list(z[f"pascal.{tag}.log"])


def test_reference_command_lines_parse():
    a = PT.parse_args(["--EVAL_DATASET", "ap10k", "--AP10K_EVAL_SUBSET", "cross-family", "--BZ", "4", "--PAIR_AUGMENT", "--WD", "0.01"])
    assert (a.AP10K_EVAL_SUBSET, a.BZ, a.PAIR_AUGMENT, a.WD, a.DATA_DIR, a.SCHEDULER) == ("cross-family", 4, True, 0.01, None, None)
