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
