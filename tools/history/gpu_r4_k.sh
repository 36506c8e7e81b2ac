#!/bin/bash
# round 4, call K: 3x3 convolutions in the 256x256 ping-pong kernel - parity, SD1.5 tower A/B (noconv5 = every conv on the 128x128 kernel),
# then the headline bench with the micro-benchmarks on the forward's own operands
O=gpurun_out/r4k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
timeout 900 python -m pytest tests/test_gpu_sd.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
for r in 1 2; do
  for v in default noconv5; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | grep -v "^weights" | tr '\n' ' ' | sed "s/^/$v $r: /"; echo
  done
done
unset VISREP_LIB
timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_forward"], {k.split()[0]: (v["ms"], v["tflops"]) for k, v in d["roofline"]["kernels"].items()})
PY
