#!/bin/bash
# round 4, call G: two-pass pre-scaled attention (fast pass without running maximum): tests + bench A/B against the committed library
O=gpurun_out/r4g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
for r in 1 2; do
  for v in default prev; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench_${v}_$r.json
    python - <<PY
import json
d=json.load(open("$O/bench_${v}_$r.json"))
print("$v $r", d["value"], d["ms_per_step"], {k.split()[0]: v["ms"] for k, v in d["roofline"]["kernels"].items()})
PY
  done
done
