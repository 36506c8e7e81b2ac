# rocprofv3 --kernel-trace of the A-score / C-score kernels at the SURVEY §8(d) sizes (profiles/round1_scores_kernel_stats.md)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/scores
python $R/tools/score_bench.py > $R/gpurun_out/scores/score_bench.json 2> $R/gpurun_out/scores/score_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/scores/prof -o p -- python $R/tools/score_bench.py > /dev/null 2>&1
python - <<PY
import sqlite3
con = sqlite3.connect("$R/gpurun_out/scores/prof/p_results.db")
rows = con.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc limit 12").fetchall()
print("kernel | calls | total ms | avg us | min us | max us")
for n, c, t, a, lo, hi in rows:
    print(n[:90].ljust(90), str(c).rjust(5), "%9.2f %9.1f %9.1f %9.1f" % (t, a, lo, hi))
PY
cat $R/gpurun_out/scores/score_bench.json
