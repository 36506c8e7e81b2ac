# same-box A/B of conv3x3_halo's tile shape: 16 x 16 pixels (one workgroup per CU) against 16 x 8 (two), kernel probe + SD1.5 tower at the sweep's launch shape
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in 16 8; do echo "== VISREP_HALO_TILE=$t"; VISREP_HALO_TILE=$t timeout 200 python tools/conv_halo_probe.py 2>&1 | grep halo; done
for r in 1 2; do for t in 16 8; do
  echo "== VISREP_HALO_TILE=$t (round $r)"; VISREP_HALO_TILE=$t timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | tail -1
done; done
