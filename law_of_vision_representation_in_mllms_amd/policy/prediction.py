"""Leave-k-out search of the AC policy — the script form of policy/prediction.py:22-79.

For every training subset of the 13 encoder settings (2 .. 13 of them) and every benchmark: fit the degree-2 policy on the
subset, predict the held-out settings, and keep the (benchmark, subset) pairs whose predicted best held-out setting is the truly
optimal one; the hits go to a csv with the reference's columns.  The table path and the output path are arguments (the reference
hard-wires a path on the author's laptop and writes into the working directory).

    python -m law_of_vision_representation_in_mllms_amd.policy.prediction ablations_t.csv [out.csv] [--max-train K]
"""
import argparse

from . import fit as F

COLUMNS = ['Benchmark', 'Train Models', 'Test MSE', 'Train MSE']


def run(table, max_train=None, benchmarks=F.BENCHMARKS, verbose=True, sizes=None):
    """[[benchmark, train subset, test MSE, train MSE], ...] in the reference's loop order (subset size, subset, benchmark).
    sizes: explicit list of training-subset sizes (default 2 .. max_train, max_train = 13)."""
    rows = []
    top = len(F.ALL_MODELS) if max_train is None else min(max_train, len(F.ALL_MODELS))
    for k in (range(2, top + 1) if sizes is None else sizes):
        for b, subset, test_mse, train_mse in F.search(table, k, benchmarks):
            if verbose:
                print(b)
                print(F.OPTIMAL[b])
            rows.append([b, subset, test_mse, train_mse])
    return rows


def main(argv=None):
    ap = argparse.ArgumentParser(description="leave-k-out search of the AC policy")
    ap.add_argument("table", help="csv with the columns of the reference's ablations_t.csv")
    ap.add_argument("out", nargs="?", default="benchmark_train_model_performance_all.csv")
    ap.add_argument("--max-train", type=int, default=None, help="largest training-subset size (default: all 13)")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    import pandas as pd
    rows = run(F.load_table(args.table), args.max_train, verbose=not args.quiet)
    pd.DataFrame(rows, columns=COLUMNS).to_csv(args.out, index=False)
    return rows


if __name__ == "__main__":
    main()
