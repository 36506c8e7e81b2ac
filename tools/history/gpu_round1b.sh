cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout=300 -x > gpurun_out/pytest2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest2.log
tail -15 gpurun_out/pytest2.log
timeout 600 python bench.py --steps 3 --warmup 1 --gemm-variant 1 --no-cpu-baseline > gpurun_out/bench2_v1.log 2>&1
echo "bench v1 rc=$?" >> gpurun_out/bench2_v1.log
timeout 600 python bench.py --steps 3 --warmup 1 --gemm-variant 2 > gpurun_out/bench2_v2.log 2>&1
echo "bench v2 rc=$?" >> gpurun_out/bench2_v2.log
tail -2 gpurun_out/bench2_v1.log; tail -2 gpurun_out/bench2_v2.log
