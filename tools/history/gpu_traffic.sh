# HBM traffic of the dominant kernels per launch: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md
# "HBM": they do not fit one pass), plus the kernel-trace stats of the same command.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/traffic/trace -- python $R/tools/gemm_probe.py 2 2 > $R/gpurun_out/traffic/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/traffic/fetch -- python $R/tools/gemm_probe.py 2 2 > $R/gpurun_out/traffic/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic/write -- python $R/tools/gemm_probe.py 2 2 > $R/gpurun_out/traffic/write.log 2>&1
cd $R; find gpurun_out/traffic -name "*.csv" | head; tail -2 gpurun_out/traffic/write.log
