R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3split3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/f32_probe.py 64 > $O/probe.log 2>&1
cd $R; python tools/summarize_pmc.py gpurun_out/r3split3 "." 2>&1 | head -24
