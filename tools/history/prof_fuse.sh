# Same-box A/B of the LayerNorm fold: rocprofv3 --kernel-trace of the bench with VISREP_FUSE_LN=0 and =1, per-kernel totals from the
# rocpd database (profiles/round1_fold_stats.md).  Run through gpurun: `gpurun -- 'bash tools/prof_fuse.sh'`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  VISREP_FUSE_LN=$f rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fuse$f -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import sqlite3
con = sqlite3.connect("$R/gpurun_out/fuse$f/p_results.db")
rows = con.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc limit 16").fetchall()
print("== VISREP_FUSE_LN=$f   kernel | calls | total ms | avg us")
for n, c, t, a in rows:
    print(n[:75].ljust(75), str(c).rjust(5), "%9.2f %9.1f" % (t, a))
PY
done
