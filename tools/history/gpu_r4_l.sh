#!/bin/bash
# round 4, call L: kernel trace of the SD1.5 tower (768 px, batch 16) after the convolutions moved into the 256x256 kernel + re-run of the conv parity test
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sd.py -m gpu -q -x --tb=short -k "conv3x3" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/sd_bench.py 16 2 768 > $O/prof.log 2>&1; echo "prof rc=$?" >> $O/prof.log
grep -v "^W2026\|amdgpu.ids" $O/prof.log | tail -4
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  {100*float(r["TotalDurationNs"])/tot:5.2f} %')
PY
find $O/prof -name "*kernel_trace.csv" -size +30M -delete
