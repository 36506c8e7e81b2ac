R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3thr
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sd.py tests/test_gpu_fullsize.py -q -x --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 200 python tools/attn_time.py 2>&1 | grep -v amdgpu | tail -4
timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores > $O/bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3thr/bench.json") if l.startswith("{")][0])
print(d["value"], d["ms_per_step"], {k:(v["ms"], v.get("tflops")) for k,v in d["roofline"]["kernels"].items()})
PY
