#!/bin/bash
# Round 3: parity of the two-blocks-per-wave attention kernel + its timing against attn_fwd<1> and the diagnostic builds.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp VISREP_DEBUG=1
P=law_of_vision_representation_in_mllms_amd
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" > gpurun_out/r3a/pytest_attn.log 2>&1; echo "pytest attention rc=$?" | tee gpurun_out/r3a/summary.txt
tail -3 gpurun_out/r3a/pytest_attn.log
timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3a/summary.txt
for v in "$@"; do
  VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so ATTN_VARIANTS=2 timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3a/summary.txt
done
