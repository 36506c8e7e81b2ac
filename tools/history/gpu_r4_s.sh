#!/bin/bash
# round 4, call S: packed C-score kernel with the next 32-channel chunk prefetched into a second register set - parity + timing against the previous library
O=gpurun_out/r4s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
timeout 900 python -m pytest tests/test_gpu_scores.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
for r in 1 2; do
  for v in default csold; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    echo "== $v $r"; timeout 300 python tools/cscore_time.py 2>&1 | grep "P="
  done
done
