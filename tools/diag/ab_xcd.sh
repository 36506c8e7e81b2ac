# same-box A/B of the XCD-weighted tile split (VISREP_XCD_BALANCE=1; 0 = equal shares of the tile list, the default) on the bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5p}; mkdir -p $O
for r in 1 2; do for h in 0 1; do
  VISREP_XCD_BALANCE=$h timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 > $O/ab_xcd.$h.$r.json 2> $O/ab_xcd.$h.$r.err
  python - $O/ab_xcd.$h.$r.json $h $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
k = {n.split()[0]: v["ms"] for n, v in r["kernels"].items()}
print(f"VISREP_XCD_BALANCE={sys.argv[2]} round {sys.argv[3]}: {d['value']} images/s, {d['ms_per_step']} ms/step, fc1 frac {r['frac']} b2b {r['frac_back_to_back']}, kernels {k}, balance {r.get('xcd_balance', {}).get('rel')} updates {r.get('xcd_balance', {}).get('updates')}")
PY
done; done
