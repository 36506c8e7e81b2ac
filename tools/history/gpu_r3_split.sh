R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3split
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f32.py -q --tb=short -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -E "^M=|passed|failed|Error|assert" $O/pytest.log | head -40
