#!/bin/bash
# round 5, call A: A-score reference-arithmetic tests + score/sweep/f32 test files on the new defaults + short bench (no sweep)
O=gpurun_out/r5a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_scores.py tests/test_gpu_sweep.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
timeout 300 python bench.py --sweep off --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
