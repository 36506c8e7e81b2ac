"""`validate_run(benchmark, train_models, top=1)` with the reference's signature (policy/validate_run.py:10-54): fit the policy on
`train_models`, rank all 13 settings by predicted performance, report whether the truly optimal one is among the `top` picks.
The table comes from `table=` / `csv=` or the VISREP_POLICY_TABLE environment variable (the reference hard-wires a laptop path)."""
import os

from . import fit as F


def validate_run(benchmark, train_models, top=1, table=None, csv=None):
    if table is None:
        path = csv or os.environ.get("VISREP_POLICY_TABLE")
        if not path:
            raise ValueError("validate_run needs the score table: pass table= / csv= or set VISREP_POLICY_TABLE")
        table = F.load_table(path)
    return F.validate_run(table, benchmark, train_models, top)
