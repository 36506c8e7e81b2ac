"""The reference-precision (fp32, split-bf16) tower forward at the sweep's chunk size, a few eager chunks: the target of
`rocprofv3 --kernel-trace --stats -- python tools/f32_forward_trace.py [products] [images]` - which kernels hold the x6 / x3 C legs of the sweep."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from law_of_vision_representation_in_mllms_amd import engine
from law_of_vision_representation_in_mllms_amd import vit_weights as VW
products = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 113
dev = torch.device("cuda", 0)
spec = VW.SPECS[bench.MODEL]
os.environ["VISREP_FAST_SYNTHETIC"] = "cuda"
eng = engine.VitEngineF32(spec, VW.synthetic_weights(spec, seed=1, n_layers=bench.N_LAYERS), dev, products=products)
px = torch.randn(n, 3, spec.image_size, spec.image_size, device=dev)
for _ in range(2): eng.forward(px, n_layers=bench.N_LAYERS)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): eng.forward(px, n_layers=bench.N_LAYERS)
e1.record(); torch.cuda.synchronize()
print(f"fp32 tower x{products}: {e0.elapsed_time(e1) / 3:.2f} ms per {n} images ({n / (e0.elapsed_time(e1) / 3e3):.1f} images/s; chunk {eng.chunk()})")
