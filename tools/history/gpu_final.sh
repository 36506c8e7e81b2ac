# round-end evidence: full GPU suite, smoke(), default bench line, rocprofv3 kernel-trace of the same bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout=300 > gpurun_out/final/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/final/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/final/smoke.log
timeout 900 python bench.py > gpurun_out/final/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/final/bench.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/prof.log 2>&1
echo "prof rc=$?" >> $R/gpurun_out/final/prof.log
cd $R
tail -3 gpurun_out/final/pytest.log; tail -2 gpurun_out/final/smoke.log; tail -2 gpurun_out/final/bench.log | cut -c1-400; find gpurun_out/final/prof -name "*.csv" | head
