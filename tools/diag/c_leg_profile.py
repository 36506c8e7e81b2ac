"""Where does one C leg's wall-clock go?  Tower pass vs exchange bookkeeping vs PCK evaluation, for one ViT setting (reference precision)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from law_of_vision_representation_in_mllms_amd import sweep as S
name = sys.argv[1] if len(sys.argv) > 1 else "CLIP224"
st = [s for s in S.SETTINGS if s.name == name][0]
dev = torch.device("cuda:0")
spair = S.synthetic_spair()
model = S.SettingModel(st, dev, precision="reference", fp32_products=3)
model.warm(S.launch_shapes(st, 100, spair, 0, 1, True, True))
px = S.ResidentPixels(dev, torch.float32)
px.prefetch(S.c_item_ids(spair, 0, 1), st.size)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); S.c_score_of(model, spair, px, dev, 0, 1); torch.cuda.synchronize(); print("c_score_of", round(time.perf_counter() - t0, 3), flush=True)
# tower pass alone
ids = S.c_item_ids(spair, 0, 1)
t0 = time.perf_counter()
for s in range(0, len(ids), st.batch):
    model.c_tokens(px(ids[s:s + st.batch], st.size))
torch.cuda.synchronize(); print("tower pass alone", round(time.perf_counter() - t0, 3))
pr = cProfile.Profile(); pr.enable(); S.c_score_of(model, spair, px, dev, 0, 1); torch.cuda.synchronize(); pr.disable()
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(28); print(buf.getvalue()[:5000])
