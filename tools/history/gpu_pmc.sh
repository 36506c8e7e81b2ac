cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_avail.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc/trace -- python $R/tools/gemm_probe.py 3 > $R/gpurun_out/pmc/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc/a -- python $R/tools/gemm_probe.py 1 > $R/gpurun_out/pmc/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc/b -- python $R/tools/gemm_probe.py 1 > $R/gpurun_out/pmc/b.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/c -- python $R/tools/gemm_probe.py 1 > $R/gpurun_out/pmc/c.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc/d -- python $R/tools/gemm_probe.py 1 > $R/gpurun_out/pmc/d.log 2>&1
cd $R; find gpurun_out/pmc -name "*.csv" | head -20; du -sh gpurun_out/pmc; tail -3 gpurun_out/pmc/a.log
