"""Two forwards in flight (round 6 experiment): the headline forward (CLIP-L/14-336, 256 images, 23 layers) stepped on ONE stream, and the same
steps alternating over TWO streams (an engine, a workspace, an output buffer and a split-K scratch per stream).  A persistent GEMM launch ends
when its slowest XCD does (the mean XCD idles 3-5 % of every launch, profiles/round5_gemm.md section 3) and the 128x128 tail pairs / split-K
reductions / statistics kernels leave most CUs idle (~5 % of the forward): a second, independent forward has work for those CUs."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from law_of_vision_representation_in_mllms_amd import _lib, engine  # noqa: E402
from law_of_vision_representation_in_mllms_amd import vit_weights as VW  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
lib = _lib.load()
spec = VW.SPECS[bench.MODEL]
w = VW.synthetic_weights(spec, seed=1, n_layers=bench.N_LAYERS)
B = bench.BATCH
NS = int(os.environ.get("STREAMS", "2"))
engs = [engine.VitEngine(spec, w, dev) for _ in range(NS)]
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
scratch = [torch.empty(256 << 20, dtype=torch.uint8, device=dev) for _ in range(NS)]
for s, buf in zip(streams, scratch):
    _lib.check(lib.visrep_set_stream_scratch(s.cuda_stream, _lib.ptr(buf), buf.numel()), "set_stream_scratch")
px = [torch.from_numpy(np.random.RandomState(2 + i).standard_normal((B, 3, spec.image_size, spec.image_size)).astype(np.float32)).to(torch.bfloat16).to(dev) for i in range(NS)]
outs = [torch.empty(B, spec.tokens, spec.d, dtype=torch.bfloat16, device=dev) for _ in range(NS)]


def run(n, nstreams):
    for i in range(n):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            engs[k].forward(px[k], n_layers=bench.N_LAYERS, out=outs[k])


def timed(nstreams):
    run(2 * nstreams, nstreams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, nstreams)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * steps / dt, dt / steps * 1e3


ref = None
for rnd in range(2):
    for ns in range(1, NS + 1):
        v, ms = timed(ns)
        print(f"round {rnd}: {ns} stream(s) in flight: {v:.1f} images/s, {ms:.3f} ms per step", flush=True)
    if ref is None:
        run(1, 1)
        torch.cuda.synchronize()
        ref = outs[0].clone()
# results do not depend on the concurrency
run(NS, NS)
torch.cuda.synchronize()
run(1, 1)
torch.cuda.synchronize()
assert torch.equal(outs[0], ref), "stream 0's output changed under concurrency"
print("outputs equal")
