#!/bin/bash
# round 4, call E: full GPU suite after the C-score packing / all_to_all / VAE-9216 test, and the bench line's scores object
O=gpurun_out/r4e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -12 $O/pytest.log
timeout 600 python bench.py --sweep off --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], {k.split()[0]: v for k, v in d["roofline"]["kernels"].items()})
print(json.dumps(d["scores"], indent=1))
PY
