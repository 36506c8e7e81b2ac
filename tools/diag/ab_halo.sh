# same-box A/B of the fused halo convolution on the SD1.5 tower at the sweep's launch shape (768 px, 16 per launch)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/gpu.sh r5g "tests:conv3x3_halo or fused_halo"
for r in 1 2; do for h in 0 1; do
  echo "== VISREP_CONV_HALO=$h (round $r)"; VISREP_CONV_HALO=$h timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | tail -1
done; done
bash tools/gpu.sh r5g "trace:tools/diag/sd_trace.py 16 768" | head -3
python - <<'PY'
import sqlite3, glob
db=glob.glob('gpurun_out/r5g/1.trace/*/*_results.db')[0]
con=sqlite3.connect(db)
for r in con.execute("select name,total_calls,average from top_kernels where name like '%halo%' or name like '%groupnorm%'"): print(r[0][:90], r[1], round(r[2],1))
PY
