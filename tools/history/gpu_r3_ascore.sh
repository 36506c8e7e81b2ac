# A-score tile variants (parity + timing), batched device pre-processing (parity) and the input-pipeline bench with it.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3as
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_scores.py tests/test_gpu_dropin.py tests/test_gpu_jpeg.py -q -x --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 300 python tools/ascore_time.py 256 2>&1 | grep -v amdgpu | tee $O/time.txt
timeout 400 python tools/pipeline_bench.py 1024 DINOv2 bf16 > $O/pipeline_bf16.json 2> $O/pipeline.err; echo "pipeline rc=$?"
cat $O/pipeline_bf16.json | python -c "import json,sys; d=json.load(sys.stdin); [print(k, v) for k, v in d.items()]"
