"""The headline forward alone (CLIP-L/14-336, 256 images, 23 layers), a few eager steps: the target of
`rocprofv3 --kernel-trace --stats -- python tools/forward_trace.py` - every kernel of one forward with its share, and (sum of kernel
time) / (wall per step) = what launch gaps cost.  Prints the wall-clock per step measured with HIP events."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from law_of_vision_representation_in_mllms_amd import engine
from law_of_vision_representation_in_mllms_amd import vit_weights as VW
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
spec = VW.SPECS[bench.MODEL]
eng = engine.VitEngine(spec, VW.synthetic_weights(spec, seed=1, n_layers=bench.N_LAYERS), dev)
B = bench.BATCH
px = torch.from_numpy(np.random.RandomState(2).standard_normal((B, 3, spec.image_size, spec.image_size)).astype(np.float32)).to(torch.bfloat16).to(dev)
out = torch.empty(B, spec.tokens, spec.d, dtype=torch.bfloat16, device=dev)
for _ in range(2): eng.forward(px, n_layers=bench.N_LAYERS, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps): eng.forward(px, n_layers=bench.N_LAYERS, out=out)
e1.record(); torch.cuda.synchronize()
print(f"forward: {e0.elapsed_time(e1) / steps:.3f} ms per step ({steps} steps, {2 + steps} forwards in the trace)")
