#!/bin/bash
# round 4, call B: full GPU suite (no -x), epilogue-addressing A/B on the bench line, full sweep in reference mode
O=gpurun_out/r4b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
for r in 1 2; do
  for v in default oldepi; do
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
timeout 900 python -m law_of_vision_representation_in_mllms_amd.sweep --also-bf16 > $O/sweep_reference.log 2>&1; tail -1 $O/sweep_reference.log > $O/sweep_reference.json
python - <<PY
import json
d=json.load(open("$O/sweep_reference.json"))
print("sweep wall", d["wall_s"], "all-bf16", d.get("wall_s_all_bf16"), "setup", d["setup_s"])
for k,v in d["per_setting"].items(): print(k, v.get("a_s"), v.get("c_s"), v.get("c_s_bf16"), v.get("dtype"), v.get("A"), v.get("pck"), v.get("pck_bf16"))
PY
