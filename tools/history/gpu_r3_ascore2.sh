R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3as2
mkdir -p $O
cd $R
for k in 0 1; do echo "FULLK=$k"; VISREP_ASCORE_FULLK=$k timeout 300 python tools/ascore_time.py 256 2>&1 | grep -v amdgpu | tee $O/time_fullk$k.txt; done
VISREP_ASCORE_FULLK=1 timeout 300 python -m pytest tests/test_gpu_scores.py -q -x --tb=short -p no:cacheprovider -k ascore 2>&1 | tail -3
timeout 300 python tools/pipeline_diag.py 512 2>&1 | grep -v amdgpu | tee $O/diag.txt
