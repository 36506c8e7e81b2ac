#!/bin/bash
# round 4, call J: kernel trace of the headline forward alone (where do the 86 ms go: tails, small kernels, gaps)
O=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/forward_trace.py 3 > $O/prof.log 2>&1; echo "prof rc=$?" >> $O/prof.log
tail -3 $O/prof.log
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over 5 forwards = {tot/5e6:.2f} ms per forward")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} us  per-forward {float(r["TotalDurationNs"])/5e6:7.3f} ms  {100*float(r["TotalDurationNs"])/tot:5.2f} %')
PY
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
