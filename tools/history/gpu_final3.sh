# round-3 end evidence in one gpurun call (every step time-bounded): full GPU suite, smoke(), the default bench line (full-size sweep, score
# kernels, fp32 tower, CPU baseline), the same bench under torch.distributed.run (world 1), rocprofv3 kernel-trace stats of the bench command
# (headline kernels + score kernels + fp32 routes), PMC passes (separate invocations, never combined with tracing) over the GEMM probe and
# the A-score kernels, and the input-pipeline bench.  Outputs under gpurun_out/final_r3/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout=500 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.log
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --sweep off --no-cpu-baseline --no-scores > $O/bench_torchrun.log 2>&1; echo "torchrun rc=$?" >> $O/bench_torchrun.log
timeout 400 python tools/pipeline_bench.py 1024 DINOv2 bf16 > $O/pipeline_bf16.json 2> $O/pipeline.err
timeout 400 python tools/pipeline_bench.py 1024 DINOv2 fp32 > $O/pipeline_fp32.json 2>> $O/pipeline.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline --sweep off > $O/prof.log 2>&1; echo "prof rc=$?" >> $O/prof.log
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_a -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_c -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_c.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_d -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_d.log 2>&1
ASCORE_SHAPES=2 timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_e -- python $R/tools/ascore_time.py 256 > $O/pmc_e.log 2>&1
ASCORE_SHAPES=2 timeout 200 rocprofv3 --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_f -- python $R/tools/ascore_time.py 256 > $O/pmc_f.log 2>&1
cd $R
tail -2 $O/pytest.log; tail -2 $O/smoke.log; tail -2 $O/bench.log | cut -c1-400; tail -2 $O/bench_torchrun.log | cut -c1-200; tail -1 $O/prof.log
python tools/summarize_pmc.py gpurun_out/final_r3 > $O/summary.md 2>&1; wc -l $O/summary.md
find $O -name "*.csv" | xargs ls -la 2>/dev/null | awk '{print $5, $9}' | head -30; du -sh $O
