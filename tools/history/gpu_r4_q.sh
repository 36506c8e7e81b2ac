#!/bin/bash
# round 4, call Q: start stagger of the persistent GEMM blocks (V5_STAGGER_P x V5_STAGGER_N sleeps of ~3.9 us) - does de-phasing the CUs' output bursts pay on v5?
O=gpurun_out/r4q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
for r in 1 2; do
  for v in default stag2x4 stag4x2 stag8x1; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench_${v}_$r.json
    python - <<PY
import json
d=json.load(open("$O/bench_${v}_$r.json"))
print("$v $r", d["value"], d["ms_per_step"], {k.split()[0]: (v["ms"], v.get("ms_in_layer_mix")) for k, v in d["roofline"]["kernels"].items()})
PY
  done
done
