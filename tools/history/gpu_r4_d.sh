#!/bin/bash
# round 4, call D: does retiring the EPI_F32X residual loads (no K-loop header wait) help or hurt the split GEMMs?
O=gpurun_out/r4d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
for r in 1 2; do
for v in default f32xnoretire; do
  if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
  echo "== $v $r"; timeout 600 python tools/f32_probe.py 64 2>&1 | grep -E "x3|x6|tower" | grep -v native
done; done
