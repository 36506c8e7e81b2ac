R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3as3
mkdir -p $O
cd $R
for k in 0 1; do echo "DEEP=$k"; VISREP_ASCORE_DEEP=$k timeout 300 python tools/ascore_time.py 256 2>&1 | grep -v amdgpu | tee $O/time_deep$k.txt; done
timeout 300 python -m pytest tests/test_gpu_scores.py -q -x --tb=short -p no:cacheprovider -k ascore 2>&1 | tail -3
timeout 300 python tools/pipeline_diag.py 512 2>&1 | grep -v amdgpu | tee $O/diag.txt
cd /tmp && export TMPDIR=/tmp
ASCORE_SHAPES=1 timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE --output-format csv -d $O/pmc_c -- python $R/tools/ascore_time.py 256 > $O/pmc_c.log 2>&1
ASCORE_SHAPES=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_a -- python $R/tools/ascore_time.py 256 > $O/pmc_a.log 2>&1
cd $R; python tools/summarize_pmc.py gpurun_out/r3as3 "ascore" 2>&1 | head -30
