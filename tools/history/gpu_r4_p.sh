#!/bin/bash
# round 4, call P: residual-tile prefetch (one dword per lane, two K-tiles ahead) in the 128x128 kernel's EPI_RESID launches: parity + SD1.5 tower A/B
O=gpurun_out/r4p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
timeout 900 python -m pytest tests/test_gpu_sd.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "conv3x3 or gemm or resid" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
for r in 1 2; do
  for v in default notouch; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | grep -v "^weights\|amdgpu.ids" | tr '\n' ' ' | sed "s/^/$v $r: /"; echo
  done
done
