#!/bin/bash
# round 4, call V: first MFMA segment of a tile with a zero C operand instead of cleared accumulators (V5_ZERO_C) - parity + bench A/B against the previous library
O=gpurun_out/r4v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sd.py tests/test_gpu_f32.py -m gpu -q -x --tb=short -k "gemm or tower or vit or conv3x3 or split" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-200
for r in 1 2 3; do
  for v in default prevz; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $r', d['value'], d['ms_per_step'], {k.split()[0]: v.get('ms_in_layer_mix') for k, v in d['roofline']['kernels'].items()})"
  done
done
