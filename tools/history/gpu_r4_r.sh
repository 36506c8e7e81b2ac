#!/bin/bash
# round 4, call R: the sweep with tile-round-aware launch / chunk sizes (ViT legs) and 32 images per diffusion launch; fp32 tower tests
O=gpurun_out/r4r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_f32.py tests/test_gpu_sweep.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
timeout 1200 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open("$O/bench.log") if x.startswith("{")][-1]
d=json.loads(l)
sw=d["sweep"]; print(d["value"], {k:v for k,v in sw.items() if k!="per_setting"})
for k,v in sw["per_setting"].items(): print(" ",k, v.get("setup_s"), v.get("a_s"), v.get("c_s"), v.get("c_s_bf16"))
print(d["scores"]["fp32_tower"])
PY
