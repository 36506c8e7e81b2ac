cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  VISREP_FUSE_LN=$f rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fuse$f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/fuse$f/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("== fuse $f")
for r in rows[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.1f"%(float(r["TotalDurationNs"])/1e6)).rjust(9), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"])
PY
done
