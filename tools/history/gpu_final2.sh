# round-2 end evidence in one gpurun call (every step time-bounded): full GPU suite, smoke(), default bench line, the same bench under
# torch.distributed.run (world 1: the launch line the driver uses for N > 1), rocprofv3 kernel-trace stats of the bench command, and the
# PMC passes over the GEMM probe (separate invocations, never combined with tracing).  Outputs under gpurun_out/final4/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout=400 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --sweep off --no-cpu-baseline > $O/bench_torchrun.log 2>&1; echo "torchrun rc=$?" >> $O/bench_torchrun.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline --sweep off > $O/prof.log 2>&1; echo "prof rc=$?" >> $O/prof.log
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_a -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_c -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_c.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_d -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_d.log 2>&1
cd $R
tail -2 $O/pytest.log; tail -2 $O/smoke.log; tail -2 $O/bench.log | cut -c1-300; tail -2 $O/bench_torchrun.log | cut -c1-200; tail -1 $O/prof.log
find $O -name "*.csv" | xargs ls -la | awk '{print $5, $9}' | head -30; du -sh $O
