# round-2 evidence in one gpurun call (every step time-bounded): rocprofv3 kernel-trace stats of the bench command, PMC passes (separate,
# never combined with tracing) over the GEMM probe and the score kernels.  Outputs under gpurun_out/r2/; summaries are copied to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2/bench -- python $R/bench.py --no-cpu-baseline --sweep off > $R/gpurun_out/r2/bench_prof.log 2>&1
echo "bench prof rc=$?"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/r2/pmc_a -- python $R/tools/gemm_probe.py 1 2 > $R/gpurun_out/r2/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/r2/pmc_c -- python $R/tools/gemm_probe.py 1 2 > $R/gpurun_out/r2/pmc_c.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r2/pmc_d -- python $R/tools/gemm_probe.py 1 2 > $R/gpurun_out/r2/pmc_d.log 2>&1
echo "gemm pmc done"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2/score_trace -- python $R/tools/score_bench.py > $R/gpurun_out/r2/score_bench.json 2> $R/gpurun_out/r2/score_trace.err
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/r2/score_pmc_a -- python $R/tools/score_bench.py > /dev/null 2> $R/gpurun_out/r2/score_pmc_a.err
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/r2/score_pmc_c -- python $R/tools/score_bench.py > /dev/null 2> $R/gpurun_out/r2/score_pmc_c.err
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/r2/score_pmc_d -- python $R/tools/score_bench.py > /dev/null 2> $R/gpurun_out/r2/score_pmc_d.err
echo "score pmc done"
cd $R
find gpurun_out/r2 -name "*.csv" | xargs ls -la | awk '{print $5, $9}' | sort -k2 | head -60
du -sh gpurun_out/r2
