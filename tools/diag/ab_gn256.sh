# same-box A/B: GroupNorm partial sums from the 256x256 convolution kernel (round 5) on the SD1.5 tower at the sweep's launch shape
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do for h in 0 1; do
  echo "== VISREP_GN_FUSE_256=$h (round $r)"; VISREP_GN_FUSE_256=$h timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | tail -1
done; done
