#!/bin/bash
# SD1.5 at 768 px, batch 16: stage times + rocprofv3 kernel trace (eager launches are what the trace lists; the graph replays the same kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3sd
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/sd_bench.py 16 3 768 > $O/sd_bench.log 2>&1; tail -3 $O/sd_bench.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/sd_bench.py 16 2 768 > $O/trace.log 2>&1
cd $R
python tools/summarize_pmc.py gpurun_out/r3sd "." > $O/summary.md 2>&1
head -40 $O/summary.md
