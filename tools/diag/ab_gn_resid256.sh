# same-box A/B (round 6): GroupNorm partial sums from the 256x256 kernel's RESIDUAL convolutions (pre-pass reads the residual tile too) on the SD1.5
# tower at the sweep's launch shape; VISREP_GN_RESID_256=0 (default) = the round-5 routing (separate statistics pass after those convolutions)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for r in 1 2; do for h in 0 1; do
  echo "== VISREP_GN_RESID_256=$h (round $r)"; VISREP_GN_RESID_256=$h timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | tail -2
done; done 2>&1 | tee gpurun_out/$1/ab_gn_resid256.log
