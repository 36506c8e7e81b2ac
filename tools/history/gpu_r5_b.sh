#!/bin/bash
# round 5, call B: the 768-px SD1.5 launch-shape test (routes + parity), bench line with the practical roof / N(0,1) entries
O=gpurun_out/r5b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sd.py -m gpu -q -x --tb=short -s -k "768px" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "routes|SD1.5 @768|passed|failed|Error|assert" $O/pytest.log | cut -c1-400 | head -20
timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5b/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], r['frac'], r['frac_back_to_back'], r['n01_back_to_back'], r['practical_roof'], r['frac_of_practical_roof'], r['whole_forward'])
PY
