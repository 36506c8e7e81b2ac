"""TFLOP/s of the implicit-GEMM 3x3 convolution at the SD1.5 VAE / UNet shapes (batch 4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import sd_engine as SE
dev = torch.device("cuda:0")
SE.ensure_scratch(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [(768, 128, 128), (384, 128, 256), (384, 256, 256), (192, 256, 512), (192, 512, 512), (96, 512, 512),
          (96, 320, 320), (48, 320, 640), (48, 640, 640), (24, 640, 1280), (24, 1280, 1280), (12, 1280, 1280), (12, 2560, 1280), (24, 1920, 1280)]
for side, ci, co in shapes:
    x = torch.randn(B * side * side, ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, 9 * ci, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(co, device=dev)
    fn = lambda: SE.conv3x3(x, B, side, side, w, b)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * side * side * co * 9 * ci
    print(f"{side:4d}^2 {ci:5d}->{co:5d}  M={B*side*side:8d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
