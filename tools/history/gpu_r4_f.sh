#!/bin/bash
# round 4, call F: pre-scaled-Q attention (accumulator-init reference): kernel + tower tests, bench A/B through VISREP_Q_PRESCALE
O=gpurun_out/r4f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_sd.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
for r in 1 2; do
  for v in 1 0; do
    VISREP_Q_PRESCALE=$v timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench_ps${v}_$r.json
    python - <<PY
import json
d=json.load(open("$O/bench_ps${v}_$r.json"))
print("prescale=$v $r", d["value"], d["ms_per_step"], {k.split()[0]: v["ms"] for k, v in d["roofline"]["kernels"].items()})
PY
  done
done
