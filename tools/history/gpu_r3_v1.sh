#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P=law_of_vision_representation_in_mllms_amd
for v in "$@"; do
  VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so ATTN_VARIANTS=1 timeout 300 python tools/attn_time.py 2>&1 | grep "attn variant"
done
ATTN_VARIANTS=1 timeout 300 python tools/attn_time.py 2>&1 | grep "attn variant"
