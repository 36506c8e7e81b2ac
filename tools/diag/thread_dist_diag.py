"""Diagnostic for tests/test_gpu_sweep.py::test_c_leg_exchange_with_device_buffers_at_world_4: where do the world-1 and the thread-emulated
world-4 runs part ways - pixels, tokens, banks after the exchange, or the evaluation?"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from law_of_vision_representation_in_mllms_amd import sweep as S
from law_of_vision_representation_in_mllms_amd.C_score import pck_train as PT
from _thread_dist import ThreadDist
DEV = "cuda:0"
P, C_ = 6, 64
spair = S.synthetic_spair(46, 120)
class Tower:
    setting = S.Setting("Elementwise", "ew", ("ew",), 12, 5)
    split = 0
    @staticmethod
    def tokens(px):
        m = px.float().reshape(px.shape[0], 3, P, 2, P, 2).mean((1, 3, 5)).reshape(px.shape[0], P * P, 1)
        f = torch.arange(1, C_ + 1, device=px.device, dtype=torch.float32).view(1, 1, C_)
        return torch.sin(m * f * 3.0 + f).to(torch.bfloat16)
pixels = lambda ids, size: S.synthetic_pixels(ids, size, DEV, torch.float32)
banks = {}
lock = threading.Lock()
orig = PT._compute_pck
def spy(args, save_path, aggre_net, files, kps, category, used_points, thresholds, bank, models, local=False):
    out = orig(args, save_path, aggre_net, files, kps, category, used_points, thresholds, bank, models, local)
    with lock:
        banks.setdefault(tag[0], {})[category] = (bank[0].clone(), out[0], out[3])
    return out
PT._compute_pck = spy
tag = ["w1a"]
a = S.c_score_of(Tower, spair, pixels, torch.device(DEV), 0, 1)
tag[0] = "w1b"
b = S.c_score_of(Tower, spair, pixels, torch.device(DEV), 0, 1)
print("world 1 twice equal:", list(a) == list(b))
for world in (2, 4):
    td = ThreadDist(world)
    S._dist = lambda: td
    PT._dist = lambda: td
    tag[0] = f"w{world}"
    got = td.run(lambda r: S.c_score_of(Tower, spair, pixels, torch.device(DEV), r, world))
    print("world", world, [list(g) == list(a) for g in got])
    for cat, (bk, pck, ic) in banks["w1a"].items():
        bk2, pck2, ic2 = banks[f"w{world}"][cat]
        same_bank = torch.equal(bk, bk2)
        if not same_bank or pck != pck2:
            nd = (bk != bk2).any(dim=-1).any(dim=-1).nonzero().flatten().tolist() if bk.shape == bk2.shape else "shape"
            print(f"  {cat}: bank equal {same_bank} (images differing: {nd}), pck {pck[:3]} vs {pck2[:3]}")
S._dist = lambda: None
PT._dist = lambda: None
# pixels / tokens drawn concurrently from four threads vs serially
ids = [ci * 100000 + i for ci in range(4) for i in range(6)]
ser = Tower.tokens(pixels(ids, 12))
res = [None] * 4
def body(r):
    res[r] = Tower.tokens(pixels(ids, 12))
th = [threading.Thread(target=body, args=(r,)) for r in range(4)]
[t.start() for t in th]; [t.join() for t in th]
print("threaded pixel draws equal the serial ones:", [torch.equal(x, ser) for x in res])
