"""How the CPU oracle (bench.py's cpu_baseline leg) scales with the thread count on the GPU box's host: seconds per ViT-L/14-336 image."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import vit_weights as VW  # noqa: E402
from oracle import vit as OV  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch default threads", torch.get_num_threads(), flush=True)
spec = VW.SPECS["openai/clip-vit-large-patch14-336"]
os.environ["VISREP_FAST_SYNTHETIC"] = "1"
w = VW.synthetic_weights(spec, seed=1, n_layers=23)
px = torch.from_numpy(np.random.RandomState(2).standard_normal((2, 3, 336, 336)).astype(np.float32))
for nt in [int(a) for a in sys.argv[1:]] or [16, 32, 64]:
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    OV.tower_features(spec, w, px[:1], select_layer=23)
    t1 = time.perf_counter()
    OV.tower_features(spec, w, px, select_layer=23)
    t2 = time.perf_counter()
    print(f"threads {nt}: first image {t1 - t0:.2f} s, then {(t2 - t1) / 2:.2f} s/image", flush=True)
