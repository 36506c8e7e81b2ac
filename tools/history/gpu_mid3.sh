# round-3 mid-round check in one gpurun call: the whole GPU suite, smoke(), and the input-pipeline bench (JSON kept for profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/mid3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout=500 --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 400 python tools/pipeline_bench.py 1024 > $O/pipeline.json 2> $O/pipeline.err; echo "pipeline rc=$?" >> $O/pipeline.err
tail -25 $O/pytest.log; tail -2 $O/smoke.log; cat $O/pipeline.json; tail -2 $O/pipeline.err
