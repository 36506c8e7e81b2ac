import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_oracle_golden import load_sd_case
from law_of_vision_representation_in_mllms_amd import sd_engine as SE
sp, wu, wv, inp, want = load_sd_case("conv_up1_ens2")
e = SE.SdEngine(sp, wu, wv, "cuda:0", up_ft_index=1, graph=False)
e.set_prompt(inp["prompt_embeds"]); e.set_timestep(261)
log = []
def wrap(name):
    fn = getattr(SE, name)
    def w(*a, **k):
        out = fn(*a, **k)
        t = out[0] if isinstance(out, tuple) else out
        torch.cuda.synchronize()
        shapes = [tuple(x.shape) for x in a if torch.is_tensor(x)]
        log.append((name, shapes, k.get("out") is not None, t.float().double().sum().item(), t.float().abs().double().sum().item()))
        return out
    setattr(SE, name, w)
for n in ("gemm", "groupnorm", "im2col3x3", "attention", "layernorm", "linear_vt", "geglu"):
    wrap(n)
torch.manual_seed(0)
lat = torch.randn(2 * 16 * 16, 8, device="cuda").to(torch.bfloat16); lat[:, 4:] = 0
runs = []
for r in range(3):
    log.clear()
    e.unet_features(lat, 2, 16, 16)
    runs.append(list(log))
bad = 0
for i, (a, b, c) in enumerate(zip(*runs)):
    if a[3:] != b[3:] or a[3:] != c[3:]:
        print(i, a[0], a[1], "inplace" if a[2] else "", a[3], b[3], c[3])
        bad += 1
        if bad > 6: break
print("ops", len(runs[0]), "first divergences shown:", bad)
