cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider --timeout=300 -k "gemm" 2>&1 | tail -4
timeout 300 python tools/gemm_msweep.py 2>&1 | grep -v amdgpu.ids | grep -E "M= 147| 65536"
