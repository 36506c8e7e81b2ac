R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/suite
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout=500 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
