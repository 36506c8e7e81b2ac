#!/bin/bash
# round 4, call T: SDXL tower at the sweep's shape (512 px, 32 per launch): stage times + kernel trace
O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/sd_bench.py 32 3 512 stabilityai/stable-diffusion-xl-base-1.0 2>&1 | grep -v amdgpu | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/sd_bench.py 32 2 512 stabilityai/stable-diffusion-xl-base-1.0 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:80]
    print(f'{n:80s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.2f} %')
PY
find $O/prof -name "*kernel_trace.csv" -size +8M -delete
