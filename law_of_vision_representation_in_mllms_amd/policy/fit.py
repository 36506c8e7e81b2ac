"""The AC policy: benchmark performance as a degree-2 polynomial of the min-max-normalised (A score, C score) - the consumer
of the two scores.  Same model as policy/fit.py:31-90, policy/validate_run.py:10-54 and the leave-k-out search
of policy/prediction.py:22-75: sklearn `PolynomialFeatures(degree=2)` + `LinearRegression()`.

This is a 13-row least-squares problem: it runs on the host in numpy (np.linalg.lstsq is what sklearn's LinearRegression calls
after centring), there is nothing for a GPU to do.  `table` is a dict of columns: 'model' (names), one column per benchmark,
'<benchmark>_average' (A score per benchmark) and 'corres' (C score) - the layout of the reference's ablations_t.csv.
"""
from __future__ import annotations

import itertools
from typing import Dict, Sequence

import numpy as np

BENCHMARKS = ["mmbench_en", "mme", "mmmu_val", "ok_vqa", "textvqa_val", "vizwiz_vqa_val", "scienceqa_img", "seed_image"]
ALL_MODELS = ["CLIP336", "CLIP224", "OpenCLIP", "DINOv2", "SDim", "SD1.5", "SDXL", "DiT", "SD3", "SD2.1", "SigLIP", "CLIP224+DINOv2", "CLIP336+DINOv2"]
OPTIMAL = {"mmbench_en": "CLIP224+DINOv2", "mme": "CLIP336", "mmmu_val": "OpenCLIP", "ok_vqa": "CLIP336+DINOv2", "textvqa_val": "CLIP336+DINOv2",
           "vizwiz_vqa_val": "CLIP336", "scienceqa_img": "CLIP336", "seed_image": "CLIP336+DINOv2"}


def load_table(path: str) -> Dict[str, np.ndarray]:
    import pandas as pd
    df = pd.read_csv(path)
    return {c: df[c].to_numpy() for c in df.columns}


def _minmax(v):
    v = np.asarray(v, np.float64)
    return (v - v.min()) / (v.max() - v.min())


def poly2(X: np.ndarray) -> np.ndarray:
    """sklearn PolynomialFeatures(degree=2) column order for 2 inputs: 1, a, c, a^2, a c, c^2 (1 input: 1, a, a^2)."""
    X = np.asarray(X, np.float64)
    cols = [np.ones(len(X))] + [X[:, i] for i in range(X.shape[1])]
    cols += [X[:, i] * X[:, j] for i in range(X.shape[1]) for j in range(i, X.shape[1])]
    return np.stack(cols, 1)


class LinearRegression:
    """Ordinary least squares with an intercept, solved like sklearn: centre X and y, minimum-norm lstsq, recover the intercept."""

    def fit(self, X, y):
        X, y = np.asarray(X, np.float64), np.asarray(y, np.float64)
        xm, ym = X.mean(0), y.mean()
        self.coef_ = np.linalg.lstsq(X - xm, y - ym, rcond=None)[0]
        self.intercept_ = ym - xm @ self.coef_
        return self

    def predict(self, X):
        return np.asarray(X, np.float64) @ self.coef_ + self.intercept_


def r2_score(y, p):
    y, p = np.asarray(y, np.float64), np.asarray(p, np.float64)
    return 1.0 - ((y - p) ** 2).sum() / ((y - y.mean()) ** 2).sum()


def normalised(table, benchmark):
    """(normed_a, normed_c, normed_y) over ALL rows of the table (fit.py:33-45 normalises with the full-table extrema)."""
    return _minmax(table[f"{benchmark}_average"]), _minmax(table["corres"]), _minmax(table[benchmark])


def fit(table, data="AC", model="polynomial", train_models: Sequence[str] = ALL_MODELS) -> Dict[str, float]:
    """policy/fit.py main loop: train R^2 per benchmark for the deterministic data choices ('AC', 'A', 'C')."""
    out = {}
    rows = np.isin(table["model"], list(train_models))
    for b in BENCHMARKS:
        a, c, y = normalised(table, b)
        X = {"AC": np.stack([a, c], 1), "A": np.stack([a, a], 1) if model == "polynomial" else a[:, None],
             "C": np.stack([c, c], 1) if model == "polynomial" else c[:, None]}[data][rows]
        if model == "polynomial":
            X = poly2(X)
        m = LinearRegression().fit(X, y[rows])
        out[b] = r2_score(y[rows], m.predict(X))
    return out


def validate_run(table, benchmark, train_models, top=1):
    """policy/validate_run.py: fit on `train_models`, rank ALL models by predicted performance, is the true optimum in the top?"""
    a, c, y = normalised(table, benchmark)
    names = np.asarray(table["model"])
    tr = np.isin(names, list(train_models))
    te = np.isin(names, ALL_MODELS)
    m = LinearRegression().fit(poly2(np.stack([a, c], 1)[tr]), y[tr])
    pred = m.predict(poly2(np.stack([a, c], 1)[te]))
    picked = list(names[te][np.argsort(pred)[-top:]])
    return OPTIMAL[benchmark] in picked, picked


def search(table, train_model_count, benchmarks=BENCHMARKS):
    """policy/prediction.py:22-75 for one subset size: every train subset, predict the best HELD-OUT model, keep the hits."""
    names = np.asarray(table["model"])
    hits = []
    for train_models in itertools.combinations(ALL_MODELS, train_model_count):
        tr = np.isin(names, train_models)
        te = np.isin(names, [m for m in ALL_MODELS if m not in train_models])
        if not tr.any() or not te.any():
            continue
        for b in benchmarks:
            a, c, y = normalised(table, b)
            P = poly2(np.stack([a, c], 1))
            m = LinearRegression().fit(P[tr], y[tr])
            pred = m.predict(P[te])
            if names[te][np.argmax(pred)] == OPTIMAL[b]:
                hits.append((b, train_models, float(((y[te] - pred) ** 2).mean()), float(((y[tr] - m.predict(P[tr])) ** 2).mean())))
    return hits
