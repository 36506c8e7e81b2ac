#!/bin/bash
# round 4, call O: C-score packed kernel with the key-point rows through LDS - parity + timing A/B
O=gpurun_out/r4o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_scores.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
timeout 300 python tools/cscore_time.py 2>&1 | grep -v amdgpu.ids | tee $O/cscore_time.log
