# same-box A/B of the fused halo convolution on the SD1.5 tower at the sweep's launch shape (768 px, 16 and 32 per launch) + SDXL at 512 px
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do for h in 0 1; do
  echo "== VISREP_CONV_HALO=$h (round $r)"; VISREP_CONV_HALO=$h timeout 300 python tools/sd_bench.py 16 3 768 2>&1 | tail -2
done; done
for h in 0 1; do echo "== batch 32 HALO=$h"; VISREP_CONV_HALO=$h timeout 300 python tools/sd_bench.py 32 3 768 2>&1 | tail -1; done
for h in 0 1; do echo "== SDXL 512 batch 32 HALO=$h"; VISREP_CONV_HALO=$h timeout 300 python tools/sd_bench.py 32 3 512 stabilityai/stable-diffusion-xl-base-1.0 2>&1 | tail -1; done
