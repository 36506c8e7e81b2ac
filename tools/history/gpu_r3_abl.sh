#!/bin/bash
# timing of the diagnostic builds of attention_ab.hip given as arguments (libvisrep_hip_<name>.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
P=law_of_vision_representation_in_mllms_amd
ATTN_VARIANTS=1,2 timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/ablate.txt
for v in "$@"; do
  VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so ATTN_VARIANTS=2 timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3a/ablate.txt
done
