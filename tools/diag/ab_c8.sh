# same-box A/B: the VAE's conv_in straight from the pixel tokens (round 5) against im2col + GEMM, SD1.5 tower at the sweep's launch shape
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do for h in 0 1; do
  echo "== VISREP_CONV_C8=$h (round $r)"; VISREP_CONV_C8=$h timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | tail -1
done; done
