#!/bin/bash
# round 4, call H: LDS-DMA issue inside the M segments (V5_DMA_IN_M=1, OWN kernels: fc1, Q|K) - correctness + bench A/B
O=gpurun_out/r4h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
VISREP_LIB=$PWD/$P/libvisrep_hip_dmainm.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "gemm or tower or vit" > $O/pytest.log 2>&1; echo "pytest(dmainm) rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
for r in 1 2; do
  for v in default dmainm; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench_${v}_$r.json
    python - <<PY
import json
d=json.load(open("$O/bench_${v}_$r.json"))
print("$v $r", d["value"], d["ms_per_step"], {k.split()[0]: v["ms"] for k, v in d["roofline"]["kernels"].items()})
PY
  done
done
