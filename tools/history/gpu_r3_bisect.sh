#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P=law_of_vision_representation_in_mllms_amd
for v in "$@"; do
  echo "== $v"
  VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "attention and attn_ab and gemm_v5" 2>&1 | tail -3
  VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so ATTN_VARIANTS=1,2 timeout 300 python tools/attn_time.py 2>&1 | grep "attn variant 2"
done
