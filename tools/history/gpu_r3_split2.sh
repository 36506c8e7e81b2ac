R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3split2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f32.py -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -E "passed|failed|Error|assert" $O/pytest.log | head -20
for a in 0 1; do echo "SPLIT_ATTENTION=$a"; VISREP_F32_SPLIT_ATTENTION=$a timeout 300 python tools/f32_probe.py 64 2>&1 | grep -E "tower|features" ; done | tee $O/probe.txt
