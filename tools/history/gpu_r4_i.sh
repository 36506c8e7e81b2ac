#!/bin/bash
# round 4, call I: image-aligned attention (attn_fwd_cls + row-mapped V GEMMs) - parity, kernel time, bench A/B (VISREP_Q_PRESCALE 1 vs 2)
O=gpurun_out/r4i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropin.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python tools/attn_time.py 2>&1 | tee $O/attn_time.log
for r in 1 2; do
  for v in 1 2; do
    VISREP_Q_PRESCALE=$v timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench_q${v}_$r.json
    python - <<PY
import json
d=json.load(open("$O/bench_q${v}_$r.json"))
print("q_mode $v run $r", d["value"], d["ms_per_step"], {k.split()[0]: v["ms"] for k, v in d["roofline"]["kernels"].items()})
PY
  done
done
