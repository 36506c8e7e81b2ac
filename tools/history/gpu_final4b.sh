# round-4 end evidence, second edition (after the image-aligned attention, the convolutions in the 256x256 kernel, the wide-head VAE attention and
# the GroupNorm work), one gpurun call, every step time-bounded.  Outputs under gpurun_out/final_r4b/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r4b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1200 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.log
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --sweep off --no-cpu-baseline --no-scores > $O/bench_torchrun.log 2>&1; echo "torchrun rc=$?" >> $O/bench_torchrun.log
timeout 300 python tools/sd_bench.py 16 3 768 > $O/sd_bench.log 2>&1
timeout 200 python tools/attn_time.py > $O/attn_time.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline --sweep off > $O/prof.log 2>&1; echo "prof rc=$?" >> $O/prof.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -- python $R/tools/forward_trace.py 3 > $O/fwd.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sd -- python $R/tools/sd_bench.py 16 2 768 > $O/sd.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_a -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_c -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_c.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_d -- python $R/tools/gemm_probe.py 1 5 > $O/pmc_d.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_g -- python $R/tools/attn_time.py > $O/pmc_g.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +16M -delete
tail -2 $O/pytest.log; tail -2 $O/smoke.log; tail -2 $O/bench.log | cut -c1-700; tail -2 $O/bench_torchrun.log | cut -c1-200; tail -1 $O/prof.log; grep -v amdgpu $O/sd_bench.log | tail -2; grep -v amdgpu $O/attn_time.log
python tools/summarize_pmc.py gpurun_out/final_r4b "gemm_bf16|attn_fwd|ascore|cscore|layernorm|splitk|ln_stats|groupnorm|softmax" > $O/summary.md 2>&1; wc -l $O/summary.md
du -sh $O
