cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/dev.log 2>&1
nproc >> gpurun_out/dev.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout=600 > gpurun_out/pytest1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest1.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench1.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench1.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
echo "prof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/prof1.log
cd $GRAFT_REPO_ROOT; find gpurun_out/prof1 -name "*.csv" | head; du -sh gpurun_out
tail -5 gpurun_out/pytest1.log; tail -3 gpurun_out/bench1.log
