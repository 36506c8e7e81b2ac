#!/bin/bash
# round 4, call M: wide-head flash attention for the VAE mid block (attn_fwd_wide<8, 4>) + row-streaming GroupNorm apply: parity, SD1.5 tower A/B
O=gpurun_out/r4m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -4 $O/pytest_attn.log
timeout 1200 python -m pytest tests/test_gpu_sd.py -m gpu -q -x --tb=short > $O/pytest_sd.log 2>&1; echo "pytest sd rc=$?"; tail -4 $O/pytest_sd.log
for r in 1 2; do
  for v in 1 0; do
    VISREP_VAE_FLASH=$v timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | grep -v "^weights\|amdgpu.ids" | tr '\n' ' ' | sed "s/^/flash=$v run $r: /"; echo
  done
done
