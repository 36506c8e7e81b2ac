#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3a
for v in 1 2 1 2; do
  timeout 600 python bench.py --sweep off --no-cpu-baseline --steps 10 --warmup 3 --attn-variant $v 2>&1 | tail -1 > gpurun_out/r3a/bench_attn$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r3a/bench_attn$v.json"))
print("attn variant $v:", d["value"], "img/s", d["ms_per_step"], "ms", {k.split()[0]: v["ms"] for k, v in d["roofline"]["kernels"].items()})
PY
done
