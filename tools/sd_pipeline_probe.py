"""Exploration for the next round: the SD1.5 tower's two halves on two HIP streams - the VAE encoder of batch i + 1 beside the UNet of batch i (the UNet is ~500
small launches that do not fill the chip; the VAE is a few dozen large ones) - against the same work on one stream.  Eager launches, independent synthetic
latents for the UNet (timing only); each stream registers its own split-K scratch (visrep_set_stream_scratch).  Usage: sd_pipeline_probe.py [batch] [iters]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, sd_engine as SE, sd_weights as SW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
sp = SW.SD_SPECS["runwayml/stable-diffusion-v1-5"]
eng = SE.SdEngine(sp, SW.synthetic_unet(sp.unet, 21, 1), SW.synthetic_vae(sp.vae, 22), dev, graph=False)
rs = np.random.RandomState(0)
img = torch.from_numpy(rs.uniform(-1, 1, (B, 3, 768, 768)).astype(np.float32)).to(dev)
eng.set_prompt(torch.from_numpy(rs.standard_normal((1, 77, 768)).astype(np.float32)))
eng.set_timestep(261)
lat = torch.randn(B * 96 * 96, 8, device=dev).to(torch.bfloat16)
lib = _lib.load()
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
scr = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(2)]
for s, b in zip((sa, sb), scr):
    _lib.check(lib.visrep_set_stream_scratch(_lib.C.c_void_p(s.cuda_stream), _lib.ptr(b), b.numel()), "scratch")
def vae(): return eng.vae_moments(img)
def unet(): return eng.unet_features(lat, B, 96, 96)
for _ in range(2): vae(); unet()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters): vae(); unet()
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / iters * 1e3
t0 = time.perf_counter()
for _ in range(iters):
    with torch.cuda.stream(sa): vae()
    with torch.cuda.stream(sb): unet()
torch.cuda.synchronize()
par = (time.perf_counter() - t0) / iters * 1e3
print(f"B={B}: one stream {seq:.1f} ms per (VAE + UNet) = {B / seq * 1e3:.1f} img/s;  two streams {par:.1f} ms = {B / par * 1e3:.1f} img/s  ({100 * (seq / par - 1):+.1f} %)")
