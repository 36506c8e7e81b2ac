#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3s
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_scores.py -x -q > gpurun_out/r3s/pytest.log 2>&1; echo "pytest sweep+scores rc=$?"; tail -3 gpurun_out/r3s/pytest.log
timeout 1500 python bench.py > gpurun_out/r3s/bench_full.json 2> gpurun_out/r3s/bench_full.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/r3s/bench_full.json
tail -5 gpurun_out/r3s/bench_full.err
