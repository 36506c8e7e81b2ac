#!/bin/bash
# round 4, call A: GPU test suite, GEMM epilogue-barrier A/B on the bench line, split-product precision + speed
O=gpurun_out/r4a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
for r in 1 2; do
  for v in default epilate; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench_${v}_$r.json
    python - <<PY
import json
d=json.load(open("$O/bench_${v}_$r.json"))
print("$v $r", d["value"], d["ms_per_step"], {k.split()[0]: v["ms"] for k, v in d["roofline"]["kernels"].items()})
PY
  done
done
unset VISREP_LIB
timeout 600 python tools/f32_probe.py 64 > $O/f32_probe.txt 2>&1; tail -25 $O/f32_probe.txt
timeout 900 python tools/precision_report.py 4 120 > $O/precision.md 2>&1; cat $O/precision.md | tail -40
