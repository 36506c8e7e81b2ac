"""AC policy fit against what the reference's own scripts compute on its data table (tests/golden/policy.npz,
produced by exec'ing policy/fit.py and policy/validate_run.py: make_golden.py gen_policy)."""
import os

import numpy as np
import pytest

from law_of_vision_representation_in_mllms_amd.policy import fit as PF

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "policy.npz"))
TABLE = {k[4:]: Z[k] for k in Z.files if k.startswith("col.")}


@pytest.mark.parametrize("data,model", [("AC", "polynomial"), ("A", "polynomial"), ("C", "linear"), ("AC", "linear")])
def test_train_r2_matches_reference_fit(data, model):
    got = PF.fit(TABLE, data, model)
    np.testing.assert_allclose([got[b] for b in PF.BENCHMARKS], Z[f"fit.{data}.{model}"], rtol=0, atol=1e-9)


def test_validate_run_matches_reference():
    for i in range(3):
        ok, picked = PF.validate_run(TABLE, str(Z[f"val.{i}.benchmark"]), [str(m) for m in Z[f"val.{i}.train"]], int(Z[f"val.{i}.top"]))
        assert ok == bool(Z[f"val.{i}.ok"]) and picked == [str(m) for m in Z[f"val.{i}.picked"]]


def test_search_finds_held_out_optimum_and_poly_features():
    assert np.array_equal(PF.poly2(np.array([[2.0, 3.0]])), [[1, 2, 3, 4, 6, 9]])
    hits = PF.search(TABLE, 12, benchmarks=["mme"])
    assert all(h[0] == "mme" and PF.OPTIMAL["mme"] not in h[1] for h in hits)
    assert PF.fit(TABLE)["mme"] > 0.95                                  # the paper's headline: R^2 of the AC law


def test_leave_k_out_search_matches_reference_script(tmp_path):
    """policy/prediction.py run as a script on the reference's table (subset sizes 2, 3, 11, 12): same hits in the same order,
    same train / test MSE; and the csv the command-line entry writes."""
    from law_of_vision_representation_in_mllms_amd.policy import prediction as PP
    rows = PP.run(TABLE, sizes=[int(k) for k in Z["pred.sizes"]], verbose=False)
    assert [r[0] for r in rows] == list(Z["pred.benchmark"])
    assert [str(tuple(r[1])) for r in rows] == list(Z["pred.train"])
    np.testing.assert_allclose([r[2] for r in rows], Z["pred.test_mse"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose([r[3] for r in rows], Z["pred.train_mse"], rtol=1e-7, atol=1e-12)
    import pandas as pd
    csv = tmp_path / "t.csv"
    pd.DataFrame({k: TABLE[k] for k in TABLE}).to_csv(csv, index=False)
    out = tmp_path / "hits.csv"
    got = PP.main([str(csv), str(out), "--max-train", "2", "--quiet"])
    back = pd.read_csv(out)
    n_pairs = sum(1 for t in Z["pred.train"] if len(eval(t)) == 2)                     # hits whose training subset has two models
    assert list(back.columns) == PP.COLUMNS and len(back) == len(got) == n_pairs > 0


def test_validate_run_module_has_the_reference_signature(tmp_path, monkeypatch):
    from law_of_vision_representation_in_mllms_amd.policy import validate_run as VR
    b, tm, top = str(Z["val.0.benchmark"]), [str(m) for m in Z["val.0.train"]], int(Z["val.0.top"])
    ok, picked = VR.validate_run(b, tm, top, table=TABLE)
    assert ok == bool(Z["val.0.ok"]) and picked == [str(m) for m in Z["val.0.picked"]]
    monkeypatch.delenv("VISREP_POLICY_TABLE", raising=False)
    with pytest.raises(ValueError):
        VR.validate_run(b, tm, top)
