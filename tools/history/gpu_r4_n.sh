#!/bin/bash
# round 4, call N: GroupNorm statistics from the producing convolution's epilogue - parity, SD tower tests, SD1.5 tower A/B (VISREP_GN_FUSE)
O=gpurun_out/r4n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_sd.py -m gpu -q -x --tb=short > $O/pytest_sd.log 2>&1; echo "pytest sd rc=$?"; tail -6 $O/pytest_sd.log | cut -c1-300
for r in 1 2; do
  for v in 1 0; do
    VISREP_GN_FUSE=$v timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | grep -v "^weights\|amdgpu.ids" | tr '\n' ' ' | sed "s/^/gn_fuse=$v run $r: /"; echo
  done
done
