"""SD1.5 tower at the sweep's launch shape, eager forwards only (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from law_of_vision_representation_in_mllms_amd import sd_engine as SE, sd_weights as SW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
side = int(sys.argv[2]) if len(sys.argv) > 2 else 768
key = sys.argv[3] if len(sys.argv) > 3 else "runwayml/stable-diffusion-v1-5"
dev = torch.device("cuda:0")
sp = SW.SD_SPECS[key]
eng = SE.SdEngine(sp, SW.synthetic_unet(sp.unet, 21, 1), SW.synthetic_vae(sp.vae, 22), dev, graph=False)
rs = np.random.RandomState(0)
img = torch.from_numpy(rs.uniform(-1, 1, (B, 3, side, side)).astype(np.float32)).to(dev)
eng.set_prompt(torch.from_numpy(rs.standard_normal((1, 77, sp.unet.cross_dim)).astype(np.float32)))
eng.set_timestep(261)
for _ in range(5):
    eng.forward(img, t=261)
torch.cuda.synchronize()
