#!/bin/bash
# round 4, call U: K = 1024 tail launches without split-K (one 128x128 launch with the epilogue fused) against the split-K pair: forward trace + bench A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r4u; mkdir -p $O
P=law_of_vision_representation_in_mllms_amd
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in default taildirect; do
    if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
    timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $r', d['value'], d['ms_per_step'])"
  done
done
export VISREP_LIB=$PWD/$P/libvisrep_hip_taildirect.so
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -- python $GRAFT_REPO_ROOT/tools/forward_trace.py 3 > $O/fwd.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/fwd -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"taildirect: total kernel time per forward {tot/5e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:60]
    print(f'{n:60s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} us  per-forward {float(r["TotalDurationNs"])/5e6:7.3f} ms')
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
