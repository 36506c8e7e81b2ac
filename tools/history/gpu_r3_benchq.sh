R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3bq
mkdir -p $O
cd $R
timeout 600 python bench.py --sweep off --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3bq/bench.json") if l.startswith("{")][0])
print(d["value"], d["ms_per_step"])
for k,v in d["scores"].items(): print(k, v)
PY
ASCORE_SHAPES=2 timeout 200 python tools/ascore_time.py 256 2>&1 | grep -v amdgpu
