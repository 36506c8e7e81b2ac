# quick loop: GPU tests + bench for the GEMM variants given as arguments (default "3")
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout=300 > gpurun_out/pytest_q.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_q.log
tail -8 gpurun_out/pytest_q.log
for v in ${@:-3}; do
  timeout 600 python bench.py --steps 3 --warmup 1 --gemm-variant $v --no-cpu-baseline > gpurun_out/bench_q$v.log 2>&1
  echo "bench v$v rc=$?" >> gpurun_out/bench_q$v.log
  tail -2 gpurun_out/bench_q$v.log
done
