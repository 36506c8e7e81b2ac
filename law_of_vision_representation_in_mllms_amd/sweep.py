"""The 13-setting A + C sweep: BASELINE.json configs[4] and the second half of its metric ("A+C score wall-clock 13 encoders").

For every vision-representation setting of the paper (policy/fit.py:20: CLIP336, CLIP224, OpenCLIP, DINOv2, SDim, SD1.5, SDXL, DiT, SD3,
SD2.1, SigLIP, CLIP224+DINOv2, CLIP336+DINOv2) the reference runs, as separate scripts with files in between,
    images -> vision tower(s) ('.'-fusion = channel concat, llava_arch.py:278-285) -> mm_projector (mlp2x_gelu, multimodal_projector/
    builder.py:40-47) -> tensor_{k}.pt -> A_score/compute.py against the CLIP336 and CLIP224 stacks, and
    SPair-71k images -> tower -> <img>_<model>.pt maps -> C_score/pck_train.py (pck_train_two.py for the two-encoder settings).
Here one launcher-aware driver does both legs per setting with every feature resident in HBM (the A features of all 13 settings are
kept: 100 x 576 x 4096 bf16 = 0.47 GB each), one process per GPU:

  image-sharded mode (default; the reference's own precedent, llava/feature/extract.py:198-214): images i = rank (mod world) for
    towers, projector and A score.  C leg: ONE tower pass over all of a setting's SPair images (every category at once, interleaved over
    the ranks, launches of equal size - `plan_launches` - so that a rank's share runs at the full launch batch whatever the world size);
    every launch's maps are all-gathered asynchronously while the next launch computes (the one real exchange step of the path), each
    rank keeps the rows of the categories it OWNS (pairs never cross categories; owners balanced by pair count) and evaluates those
    categories alone; one all_gather_object of the per-category results ends the setting.  A leg: one all-reduce of three fp64 sums per
    setting.
  encoder-sharded A score (BASELINE.json configs[2]; `a_scores_encoder_sharded`): the CLIP336 / CLIP224 reference stacks are produced
    image-sharded and all-gathered in image chunks on the collective's own stream while the next chunk's towers run; each rank then
    runs ITS OWN encoders (setting i on rank i mod world) over all images and scores them against the gathered references.

Timing: per setting, `device sync + barrier` on both sides of [features + scores]; weight creation and engine construction are outside
the timed region (as the weights of the headline bench are).  Everything numeric goes through the drop-in surfaces (tower registry,
build_vision_projector, ascore_ops, C_score.pck_train._compute_pck), so the sweep measures the product path.
"""
from __future__ import annotations

import contextlib
import os
import time
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

CLIP336, CLIP224 = 'openai/clip-vit-large-patch14-336', 'openai/clip-vit-large-patch14'
OPENCLIP, DINOV2, SIGLIP = 'laion/CLIP-ViT-L-14-laion2B-s32B-b82K', 'facebook/dinov2-large', 'google/siglip-base-patch16-224'
SD15, SD21, SDXL = 'runwayml/stable-diffusion-v1-5', 'stabilityai/stable-diffusion-2-1', 'stabilityai/stable-diffusion-xl-base-1.0'
IMSD, DIT, SD3 = 'lambdalabs/sd-image-variations-diffusers', 'facebook/DiT-XL-2-512', 'stabilityai/stable-diffusion-3-medium-diffusers'


@dataclass(frozen=True)
class Setting:
    name: str                 # policy/fit.py:20 (train_models)
    key: str                  # A_score/compute.py:10 subfolder name
    towers: tuple             # registry ids (llava_arch.py:29-40); two ids = '.'-fusion / pck_train_two
    size: int                 # input side: ViT towers their own resolution; diffusion towers C_score/extract_feature.py:56-63
    batch: int                # images per tower launch


# order of policy/fit.py:20; the two CLIP stacks come first there too, which is what the A score needs (its references).  Diffusion towers run
# 32 images per launch: SD1.5 at 768 px does 211 / 209 / 217 / 215 images/s at batch 16 / 24 / 32 / 48 (tools/sd_bench.py, end of round 4: the
# UNet's small kernels fill the chip better; 15.7 GiB peak at 32; round 1-3 code: 160 / 173 / 180 at batch 4 / 8 / 16).
# ViT launch sizes are picked for the GEMMs' tile rounds (256 x 256 tiles on 256 CUs; the N = 1024 projections have 4 column tiles) and the
# C leg issues FULL launches + one remainder (plan_launches): 113 images of a 577-token tower = 3.98 rounds of the out / V GEMMs (4 run; 128 or
# an equalised 120 would be 4.5 / 4.23: 5 run), 7.9 / 15.8 of Q|K / fc1; 256 images of a 257-token tower = 4.02 rounds (the 4-tile remainder goes
# to the tail launch; an equalised 225 would be 3.53).  The reference-precision engine cuts a launch into chunks by the same rule
# (engine.best_chunk: 113 images of a 577-token tower, 127 of a 257-token one - exactly two rounds, no tail launch).
_B224 = int(os.environ.get("VISREP_SWEEP_B224", "256"))        # A/B knob (tools/): launch size of the 257-token towers (255 = whole tile rounds, no tail)
SETTINGS = (
    Setting("CLIP336", "clip336", (CLIP336,), 336, 113),
    Setting("CLIP224", "clip224", (CLIP224,), 224, _B224),
    Setting("OpenCLIP", "openclip", (OPENCLIP,), 224, _B224),
    Setting("DINOv2", "dino", (DINOV2,), 224, _B224),
    Setting("SDim", "imsd", (IMSD,), 768, 32),
    Setting("SD1.5", "sd1.5", (SD15,), 768, 32),
    Setting("SDXL", "sdxl", (SDXL,), 512, 32),
    Setting("DiT", "dit", (DIT,), 512, 32),
    Setting("SD3", "sd3", (SD3,), 512, 32),
    Setting("SD2.1", "sd2.1", (SD21,), 768, 32),
    Setting("SigLIP", "siglip", (SIGLIP,), 224, 256),
    Setting("CLIP224+DINOv2", "clip224+dino", (CLIP224, DINOV2), 224, _B224),
    Setting("CLIP336+DINOv2", "clip336+dino", (CLIP336, DINOV2), 336, 113),
)
REFS = ("clip336", "clip224")
# The dtype the reference runs each tower in on its C path (C_score/extract_feature.py:36-50,80-91: CLIP / OpenCLIP / DINOv2 are built with no
# dtype cast and fed fp32 pixels; SigLIP and every diffusion tower run in bf16) - keyed by registry id; anything else (diffusion ids) = bf16.
# Its A path is LLaVA's: the whole model is `.to(bfloat16)`, so every tower and the projector run in bf16 there.
_C_REF_NAME = {CLIP336: "CLIP", CLIP224: "CLIP", OPENCLIP: "OPENCLIP", DINOV2: "DINOv2", SIGLIP: "SigLIP"}


def reference_c_precision(tower_id: str) -> str:
    from .C_score.extract_feature import _REF_PRECISION
    return _REF_PRECISION.get(_C_REF_NAME.get(tower_id, ""), "bf16")
SPAIR_CATEGORIES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "dog", "horse", "motorbike",
                    "person", "pottedplant", "sheep", "train", "tvmonitor")


# ------------------------------------------------------------------------------------------------ synthetic workloads
@dataclass
class SpairCategory:
    name: str
    files: List[str]          # 2 N names (source, target per pair) - distinct images are distinct names
    slot: np.ndarray          # [2 N] index of each file among the category's distinct images
    n_images: int
    kps: torch.Tensor         # [2 N, K, 3] (x, y, visible) in the 840-px annotation frame
    thresholds: np.ndarray    # [N] bbox thresholds


def synthetic_spair(n_images: int = 1800, n_pairs: int = 12234, kmax: int = 20, seed: int = 5, categories: Sequence[str] = SPAIR_CATEGORIES):
    """SPair-71k-shaped evaluation set (the dataset is absent offline): 18 categories, images reused across pairs as in SPair
    (~1,800 distinct test images for 12,234 pairs), K ~ U{3..kmax} visible key points per pair, bbox thresholds U[150, 700]."""
    rs = np.random.RandomState(seed)
    nc = len(categories)
    out = []
    for c, name in enumerate(categories):
        ni = n_images // nc + (1 if c < n_images % nc else 0)
        npair = n_pairs // nc + (1 if c < n_pairs % nc else 0)
        ni = max(ni, 2)
        src, trg = rs.randint(0, ni, npair), rs.randint(0, ni, npair)
        slot = np.stack([src, trg], 1).reshape(-1).astype(np.int32)
        kps = np.zeros((2 * npair, kmax, 3), np.float32)
        kps[:, :, :2] = rs.uniform(0, 839, (2 * npair, kmax, 2))
        nk = rs.randint(3, kmax + 1, npair)
        vis = (np.arange(kmax)[None] < nk[:, None]).astype(np.float32)
        kps[0::2, :, 2] = vis
        kps[1::2, :, 2] = vis * (rs.rand(npair, kmax) > 0.1)
        files = [f"{name}/{i:05d}.jpg" for i in slot]
        out.append(SpairCategory(name, files, slot, ni, torch.from_numpy(kps), rs.uniform(150, 700, npair)))
    return out


def synthetic_pixels(ids: Sequence[int], size: int, device, dtype=torch.bfloat16, seed: int = 0) -> torch.Tensor:
    """[len(ids), 3, size, size] in [-1, 1): one generator draw per GLOBAL image id, so a rank's share of the images is the same
    tensor whatever the world size (the sharded sweep reproduces the single-process numbers)."""
    g = torch.Generator(device=device)
    out = torch.empty(len(ids), 3, size, size, dtype=dtype, device=device)
    for j, i in enumerate(ids):
        g.manual_seed(seed * 1000003 + int(i))
        out[j] = (torch.rand(3, size, size, generator=g, device=device) * 2 - 1).to(dtype)
    return out


class ResidentPixels:
    """The sweep's default image source: every image a setting will ask for is drawn in that setting's SETUP (`prefetch`, outside the timed
    region) into one HBM tensor per input size, and a launch's batch is one index_select from it - the timed region starts with its inputs
    resident in HBM, as the bench contract words it.  (Until the end of round 4 the draw happened inside the timed legs: one generator seed +
    five tiny kernels per image, ~0.15 s per setting of pure launch overhead - 5 % of the sweep's wall-clock.)  Same images as
    `synthetic_pixels` (one generator draw per global image id), so results do not depend on when they were drawn; ids that were not
    prefetched are drawn on the spot."""

    def __init__(self, device, dtype=torch.float32, seed: int = 0):
        self.device, self.dtype, self.seed = device, dtype, seed
        self._row: Dict[int, Dict[int, int]] = {}        # size -> image id -> row
        self._store: Dict[int, torch.Tensor] = {}

    def prefetch(self, ids: Sequence[int], size: int) -> None:
        rows = self._row.setdefault(size, {})
        new = [int(i) for i in dict.fromkeys(int(i) for i in ids) if int(i) not in rows]
        if not new:
            return
        fresh = synthetic_pixels(new, size, self.device, self.dtype, self.seed)
        old = self._store.get(size)
        base = 0 if old is None else old.shape[0]
        self._store[size] = fresh if old is None else torch.cat([old, fresh], 0)
        for k, i in enumerate(new):
            rows[i] = base + k

    def __call__(self, ids: Sequence[int], size: int) -> torch.Tensor:
        rows = self._row.get(size, {})
        ids = [int(i) for i in ids]
        if any(i not in rows for i in ids):
            return synthetic_pixels(ids, size, self.device, self.dtype, self.seed)
        return self._store[size].index_select(0, torch.tensor([rows[i] for i in ids], dtype=torch.long, device=self.device))


def c_item_ids(spair, rank: int, world: int) -> List[int]:
    """The global image ids `c_score_of` asks the image source for on this rank (its share of the items, padded like the launches)."""
    items = [(ci, i) for ci, cat in enumerate(spair) for i in range(cat.n_images)]
    if not items:
        return []
    per = (len(items) + world - 1) // world
    mine = items[rank::world]
    mine = mine + [mine[-1] if mine else items[-1]] * (per - len(mine))
    return [ci * 100000 + i for ci, i in mine]


# ------------------------------------------------------------------------------------------------ one setting's towers + projector
class SettingModel:
    """Tower(s) + mlp2x_gelu projector of one setting, built through the drop-in registry (llava_arch.build_function_mapping)."""

    def __init__(self, setting: Setting, device, hidden: int = 4096, synthetic: bool = True, precision: str = "bf16", fast_weights: bool = True,
                 fp32_products: Optional[int] = None, alt_fp32_products: Optional[int] = None):
        """fp32_products: split-bf16 product set of the reference-precision (fp32) ViT engines this setting builds - None = the engine's
        fp32-equivalent default (6); 3 = the throughput set (two-plane operands), an explicit opt-in that the `dtypes` labels carry
        ("fp32[split-bf16 x3]") into the sweep's output.
        precision:
          'reference' = the reference's own arithmetic per leg: A leg in bf16 (LLaVA's model.to(bfloat16): towers and projector), C leg per
                        tower as C_score/extract_feature.py builds it (`reference_c_precision`: fp32 for CLIP / OpenCLIP / DINOv2 and both
                        fusions, bf16 for SigLIP and the diffusion towers) - a setting whose two legs differ holds two engines per tower;
          'bf16'      = MFMA throughput engines on both legs + bf16 projector;
          'fp32'      = reference-precision ViT towers and an fp32 projector on both legs (diffusion towers are bf16 in the reference too).
        fast_weights: draw the synthetic weights with the GPU's generator (seconds instead of minutes for the 1-3 G parameter diffusion
        models); False keeps the version-stable numpy stream."""
        from .llava.model import llava_arch as LA
        from .llava.model.multimodal_projector.builder import build_vision_projector
        if precision not in ("reference", "bf16", "fp32"):
            raise ValueError(f"precision must be 'reference', 'bf16' or 'fp32', got {precision!r}")
        self.setting, self.device = setting, torch.device(device)
        env = {"VISREP_SYNTHETIC_WEIGHTS": "1"} if synthetic else {}
        if synthetic and fast_weights:
            env["VISREP_FAST_SYNTHETIC"] = "cuda" if self.device.type == "cuda" else "1"

        def tower(tid, prec, products=fp32_products):
            cfg = SimpleNamespace(mm_vision_tower=tid, vision_tower=tid, mm_vision_select_layer=-2, mm_vision_select_feature='patch',
                                  up_ft_index=0, t=1, prompt='', ensemble_size=1, img_size=setting.size,        # train.py:83-87 defaults
                                  vit_img_size=setting.size, synthetic_weights=synthetic, device=self.device, tower_precision=prec,
                                  tower_products=products)
            return LA.build_function_mapping[tid](cfg)
        a_prec = "fp32" if precision == "fp32" else "bf16"
        with _environ(env):
            self.towers = [tower(tid, a_prec) for tid in setting.towers]                      # A leg
            self.c_towers = []                                                                # C leg: the same objects unless the dtype differs
            for tid, ta in zip(setting.towers, self.towers):
                c_prec = reference_c_precision(tid) if precision == "reference" else a_prec
                same = (ta.dtype == torch.float32) == (c_prec == "fp32") or hasattr(ta, "up_ft_index")
                self.c_towers.append(ta if same else tower(tid, c_prec))
            # alt_fp32_products: the fp32 C-leg towers once more with another split-bf16 product set (bench.py times the throughput set, 3,
            # beside the fp32-equivalent 6 the reference-precision leg runs); same weights (same seed), engines of their own
            self.alt_c_towers = None
            if alt_fp32_products is not None and any(t.dtype == torch.float32 for t in self.c_towers):
                self.alt_c_towers = [tower(tid, "fp32", alt_fp32_products) if tc.dtype == torch.float32 else tc for tid, tc in zip(setting.towers, self.c_towers)]
        def label(t):
            if t.dtype != torch.float32:
                return "bf16"
            eng = getattr(getattr(t, "vision_tower", None), "engine", None)
            pr = getattr(eng, "products", None)
            return f"fp32[split-bf16 x{pr}]" if pr else "fp32"                                    # pr None: exact-fp32 MFMA route
        name = lambda ts: "+".join(sorted({label(t) for t in ts}))
        self.dtypes = {"a": name(self.towers), "c": name(self.c_towers)}
        if self.alt_c_towers is not None:
            self.dtypes["c_alt"] = name(self.alt_c_towers)
        self.width = sum(t.hidden_size for t in self.towers)
        torch.manual_seed(7)                                             # same projector on every rank
        self.projector = build_vision_projector(SimpleNamespace(mm_projector_type='mlp2x_gelu', mm_hidden_size=self.width, hidden_size=hidden))
        if precision != "fp32":
            self.projector = self.projector.to(torch.bfloat16)           # what LLaVA's model.to(bfloat16) leaves: the bf16 MFMA path
        self.split = self.towers[0].hidden_size if len(self.towers) == 2 else 0
        self._px_dtype = torch.float32 if any(t.dtype == torch.float32 for t in self.towers + self.c_towers) else torch.bfloat16

    def warm(self, shapes: Sequence[int]) -> None:
        """Untimed warm-up at every launch shape the sweep will use on this rank (HIP-graph capture of the diffusion towers, engine
        workspaces), on both legs' engines: no capture or allocation is left for the timed region."""
        if self.device.type != "cuda":
            return
        for n in sorted(set(int(x) for x in shapes if x > 0)):
            px = synthetic_pixels(range(n), self.setting.size, self.device, self._px_dtype)
            self.project(self.tokens(px))
            if any(c is not a for c, a in zip(self.c_towers, self.towers)):
                self.c_tokens(px)
            if self.alt_c_towers is not None:
                self.c_tokens_alt(px)

    @staticmethod
    def _run(towers, px):
        f = [t(px if t.dtype == px.dtype else px.to(t.dtype)) for t in towers]               # every tower is fed (and answers in) its own dtype
        return f[0] if len(f) == 1 else torch.cat(f, dim=-1)

    @torch.no_grad()
    def tokens(self, px: torch.Tensor) -> torch.Tensor:
        """A leg: [B, 3, s, s] -> tower tokens [B, N, C] ('.'-fusion: channel concat of the towers' tokens, llava_arch.py:278-285)."""
        return self._run(self.towers, px)

    @torch.no_grad()
    def c_tokens(self, px: torch.Tensor) -> torch.Tensor:
        """C leg: the same towers in the dtype C_score/extract_feature.py runs them in (pck_train_two: per-encoder maps, concatenated)."""
        return self._run(self.c_towers, px)

    @torch.no_grad()
    def c_tokens_alt(self, px: torch.Tensor) -> torch.Tensor:
        """C leg on the alternative product set's engines (alt_fp32_products)."""
        return self._run(self.alt_c_towers, px)

    @torch.no_grad()
    def project(self, tok: torch.Tensor) -> torch.Tensor:
        return self.projector(tok)


@contextlib.contextmanager
def _environ(kv: Dict[str, str]):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# ------------------------------------------------------------------------------------------------ collectives helpers
def _dist():
    d = torch.distributed
    return d if d.is_available() and d.is_initialized() else None


def _fence(device):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    d = _dist()
    if d is not None:
        d.barrier()
        if device is not None and torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)


def all_gather_rows(local: torch.Tensor, n_total: int, rank: int, world: int, async_op: bool = False):
    """local = rows i = rank (mod world) of a [n_total, ...] tensor -> the full tensor in global row order on every rank.
    Rows are padded to ceil(n_total / world) per rank so the collective is one equal-size all-gather (RCCL all_gather over xGMI).
    async_op: returns (handle, finish) - the gather runs on the collective's own stream; finish() waits and reorders."""
    d = _dist()
    per = (n_total + world - 1) // world
    if d is None or world == 1:
        return (None, lambda: local) if async_op else local
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    h = d.all_gather(parts, pad, async_op=async_op)

    def finish():
        if h is not None:
            h.wait()
        stacked = torch.stack(parts, 1)                               # [per, world, ...]: row j of rank r is global row j * world + r
        return stacked.reshape((per * world,) + tuple(local.shape[1:]))[:n_total].contiguous()
    return (h, finish) if async_op else finish()


# ------------------------------------------------------------------------------------------------ A leg
def _a_hooks():
    from . import ascore_ops
    return ascore_ops.max_cos_mean, ascore_ops.row_scales


def a_features(model, n_images: int, ids: Sequence[int], pixels: Callable) -> torch.Tensor:
    """projected features [len(ids), N, D] of the images `ids` (global ids)."""
    out = []
    B = model.setting.batch
    for s in range(0, len(ids), B):
        chunk = ids[s:s + B]
        out.append(model.project(model.tokens(pixels(chunk, model.setting.size))))
    return torch.cat(out, 0) if out else None


def a_score_of(feat: torch.Tensor, refs: Dict[str, tuple], device, hooks=None) -> float:
    """A_score/compute.py:51-81 for one encoder over this rank's images (all ranks' sums are all-reduced): mean over images of
    mean_t max_s cos against clip336 and clip224, then the mean of the two."""
    score, scales = hooks or _a_hooks()
    n = 0 if feat is None else feat.shape[0]
    acc = torch.zeros(3, dtype=torch.float64)
    if n:
        fs = scales(feat)
        for k, r in enumerate(REFS):
            ref, rs = refs[r]
            s = score(feat, ref[:n], fs, rs[:n] if rs is not None else None).double().cpu()
            acc[k] = float(sum(float(s[i]) for i in range(n)))
        acc[2] = n
    d = _dist()
    if d is not None:
        acc = acc.to(device)
        d.all_reduce(acc)
        acc = acc.cpu()
    return float((acc[0] / acc[2] + acc[1] / acc[2]) / 2)


# ------------------------------------------------------------------------------------------------ C leg
def _c_args(P: int, window: int = 5):
    return SimpleNamespace(NUM_PATCHES=P, SOFT_EVAL=True, SOFT_EVAL_WINDOW=window, ANNO_SIZE=840, EVAL_DATASET='spair', TRAIN_DATASET='spair',
                           KPT_RESULT=False, ENSEMBLE=1, BBOX_THRE=True, ADAPT_FLIP=False, TOTAL_SAVE_RESULT=0, COMPUTE_GEOAWARE_METRICS=False,
                           MODEL="fused")


def plan_launches(n: int, batch: int, equal: bool = True) -> List[int]:
    """n images in ceil(n / batch) launches, at most two distinct shapes (every shape is warmed - HIP-graph captured - in setup).
    equal (diffusion towers): (almost) equal sizes, differing by one, so that a rank's share never ends in a near-empty launch (225 images
    at batch 32 -> 8 x 28 / 29, not 7 x 32 + 1).  Not equal (ViT towers): full launches + one remainder - their launch size is picked for
    the GEMMs' tile rounds (SETTINGS), which an equalised 225 instead of 256 would undo (3.5 rounds of the N = 1024 projections: 4 run)."""
    if n <= 0:
        return []
    k = (n + batch - 1) // batch
    if not equal:
        return [batch] * (n // batch) + ([n % batch] if n % batch else [])
    return [n // k + (1 if j < n % k else 0) for j in range(k)]


def category_owners(spair: Sequence[SpairCategory], world: int) -> List[int]:
    """Owner rank of every category: longest-processing-time greedy on the pair counts (ties by index), identical on every rank.
    Pairs never cross categories (C_score/pck_train.py:315-340 evaluates category by category), so an owner needs no other bank."""
    load = [0] * world
    owner = [0] * len(spair)
    for ci in sorted(range(len(spair)), key=lambda c: (-len(spair[c].thresholds), c)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[ci] = r
        load[r] += len(spair[ci].thresholds)
    return owner


def launch_shapes(setting: Setting, n_a_images: int, spair, rank: int, world: int, do_a: bool = True, do_c: bool = True) -> List[int]:
    """Every tower launch size `run_sweep` issues for this setting on this rank (A leg: the rank's images in launches of `batch`)."""
    shapes = []
    if do_a:
        n = len(range(rank, n_a_images, world))
        shapes += [min(setting.batch, n - s) for s in range(0, n, setting.batch)]
    if do_c and spair:
        shapes += c_launch_plan(spair, setting.batch, world)
    return shapes


def c_launch_plan(spair: Sequence[SpairCategory], batch: int, world: int) -> List[int]:
    """Launch sizes of ONE rank's share of a setting's C images (the same on every rank: short ranks repeat their last image).
    Launch batches above 32 are the ViT towers': full launches + a remainder (plan_launches)."""
    n_items = sum(c.n_images for c in spair)
    return plan_launches((n_items + world - 1) // world, batch, equal=batch <= 32)


def c_exchange_plan(n_items: int, item_owner: Sequence[int], world: int, off: int, sz: int):
    """Who sends which rows of one launch to whom.  Every rank holds the global items g = j * world + r at local position j (the last local
    positions of a short rank are padding: a repeated image, sent nowhere); the launch covers local positions [off, off + sz) on every rank.
    Returns send[r][q] = local positions (relative to the launch) that rank r sends to rank q, in increasing order - the same table on
    every rank, so receivers know what arrives: rows from r land in the order of send[r][me].  A row goes to exactly ONE rank (the owner of
    its category): 1 / world of an all-gather's bytes."""
    send = [[[] for _ in range(world)] for _ in range(world)]
    for r in range(world):
        for j in range(off, off + sz):
            g = j * world + r
            if g < n_items:
                send[r][item_owner[g]].append(j - off)
    return send


def c_score_of(model, spair: Sequence[SpairCategory], pixels: Callable, device, rank: int, world: int, tokens: Optional[Callable] = None,
               timing: Optional[dict] = None):
    """pck_train.eval (pck_train.py:315-340) over the synthetic SPair set, every feature resident in HBM.  tokens: the tower pass
    ([B, 3, s, s] -> [B, N, C]); default = the model's C-leg engines (`c_tokens`, else `tokens`).
    1. ONE image-sharded tower pass over all categories' distinct images (global item g = (category, image) on rank g mod world) in
       `c_launch_plan` launches; 2. each launch's maps go to the rank that OWNS their category - an owner-addressed all_to_all_single (RCCL
       all-to-all over xGMI: every row crosses the fabric once; the round-3 all-gather sent it to all `world` ranks), asynchronous: the
       exchange of launch j runs under launch j + 1; 3. the owner evaluates its categories with _compute_pck(local=True);
    4. one all_gather_object of (pck, img_correct) per category; statistics are accumulated in category order on every rank, so the
       result does not depend on the world size.
    timing: a dict that receives {"tower_s", "eval_s"} - the leg's wall-clock split at the point where this rank's banks are complete (one extra
    device sync there): what the scaling prediction (`predict_scaling`) needs, since the two parts shard differently."""
    from .C_score import pck_train as PT
    from .C_score.utils.logger import log_weighted_pcks, update_stats
    aggre = PT.DummyAggregationNetwork()
    tokens = tokens or getattr(model, "c_tokens", None) or model.tokens
    t_begin = time.perf_counter()
    d = _dist() if world > 1 else None
    items = [(ci, i) for ci, cat in enumerate(spair) for i in range(cat.n_images)]
    n_items = len(items)
    owner = category_owners(spair, world)
    per = (n_items + world - 1) // world
    mine = items[rank::world]
    mine = mine + [mine[-1] if mine else items[-1]] * (per - len(mine))      # equal shapes on every rank: a short rank repeats an image
    plan = c_launch_plan(spair, model.setting.batch, world)
    # where a row lands: global item -> row of this rank's bank storage (the categories it owns, in category order)
    base, tot = {}, 0
    for ci, cat in enumerate(spair):
        if owner[ci] == rank:
            base[ci] = tot
            tot += cat.n_images
    item_owner = [owner[ci] for ci, _ in items]
    dest_of = [base[ci] + i if ci in base else -1 for ci, i in items]
    store, inflight, off = None, None, 0

    def alloc(like):
        nonlocal store
        if store is None:
            store = torch.empty((max(tot, 1),) + tuple(like.shape[1:]), dtype=torch.float32, device=like.device)

    for sz in plan:
        chunk = mine[off:off + sz]
        tok = tokens(pixels([ci * 100000 + i for ci, i in chunk], model.setting.size)).contiguous()
        alloc(tok)
        if inflight is not None:
            inflight()
        send = c_exchange_plan(n_items, item_owner, world, off, sz)
        # rows that arrive here from rank r, in order: r's local positions send[r][rank] -> global item -> bank row
        dst = [dest_of[(off + j) * world + r] for r in range(world) for j in send[r][rank]]
        if d is None:
            if dst:
                store.index_copy_(0, torch.tensor(dst, dtype=torch.long, device=tok.device), tok.index_select(0, torch.tensor(send[0][0], dtype=torch.long, device=tok.device)).float())
            inflight = None
        else:
            order = [j for q in range(world) for j in send[rank][q]]                      # my rows grouped by destination rank
            sbuf = tok.index_select(0, torch.tensor(order, dtype=torch.long, device=tok.device)) if order else tok[:0]
            in_splits = [len(send[rank][q]) for q in range(world)]
            out_splits = [len(send[r][rank]) for r in range(world)]
            rbuf = torch.empty((sum(out_splits),) + tuple(tok.shape[1:]), dtype=tok.dtype, device=tok.device)
            h = d.all_to_all_single(rbuf, sbuf, out_splits, in_splits, async_op=True)

            def finish(h=h, rbuf=rbuf, sbuf=sbuf, dst=dst):
                h.wait()
                if dst:
                    store.index_copy_(0, torch.tensor(dst, dtype=torch.long, device=rbuf.device), rbuf.float())
            inflight = finish
        off += sz
    if inflight is not None:
        inflight()
    if store is None:
        raise ValueError("empty SPair set")
    if timing is not None:
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)
        timing["tower_s"] = time.perf_counter() - t_begin
    P = int(round(store.shape[1] ** 0.5))
    if P * P != store.shape[1]:
        raise ValueError(f"{model.setting.name}: {store.shape[1]} tokens is not a square map")
    args = _c_args(P)
    results = {}
    for ci, cat in enumerate(spair):
        if owner[ci] != rank:
            continue
        bank, layout = store[base[ci]:base[ci] + cat.n_images], "pc"
        if bank.shape[2] % 4 or model.split % 4:
            bank, layout = bank.transpose(1, 2).contiguous(), "cp"
        pck, _, _, img_correct = PT._compute_pck(args, ".", aggre, cat.files, cat.kps, cat.name, None, cat.thresholds,
                                                 (bank, cat.slot, model.split, layout), models=("fused",), local=True)
        results[ci] = (pck, img_correct)
    del store
    if d is not None:
        allr = [None] * world
        d.all_gather_object(allr, results)
        results = {k: v for part in allr for k, v in part.items()}
    pcks, pcks_05, pcks_01, weights, kpt_weights = ([] for _ in range(5))
    for ci in range(len(spair)):
        update_stats(args, pcks, pcks_05, pcks_01, weights, kpt_weights, *results[ci])
    import logging
    quiet = logging.getLogger("visrep.sweep")
    if timing is not None:
        timing["eval_s"] = time.perf_counter() - t_begin - timing["tower_s"]
    return log_weighted_pcks(args, quiet, pcks, pcks_05, pcks_01, weights)


# ------------------------------------------------------------------------------------------------ predicted scaling (no multi-GPU box so far)
VIT_LAUNCH_FIXED_IMAGES = 24       # a ViT tower launch costs about (images + 24) image-times (tile-round quantisation + launch tail)
PER_SETTING_SYNC_S = 0.025         # fences, the all_gather_object, Python between the legs


def predict_scaling(per: dict, settings: Sequence[Setting], n_a_images: int, spair, worlds=(2, 4, 8)) -> dict:
    """What the measured single-GPU legs predict for N ranks - a MODEL, printed next to the measured row so that the first multi-GPU run has
    something to be compared with; it is not a measurement.  Per setting: the C leg's tower part (`c_tower_s`) scales with the cost of one rank's
    launch plan (`c_launch_plan`; ViT launch = images + 24, diffusion launch = images), its evaluation part (`c_eval_s`) with the largest
    owner's share of the pairs (`category_owners`), the A leg with a rank's share of the A images in launches of `batch`, plus 25 ms of fences per
    setting.  NOT modelled: RCCL's all_to_all kernels competing for CUs with the persistent GEMMs that own every CU (the exchange of launch j
    runs under launch j + 1), host launch overhead of the graph replays, clock differences between GPUs, rank skew at the fences."""
    def launch_cost(plan, vit):
        return sum(x + (VIT_LAUNCH_FIXED_IMAGES if vit else 0) for x in plan)
    by_name = {s.name: s for s in settings}
    total_pairs = sum(len(c.thresholds) for c in spair)
    out = {}
    base = sum(v.get("a_s", 0.0) + v.get("c_s", 0.0) for v in per.values())
    for n in worlds:
        owners = category_owners(spair, n)
        load = [0] * n
        for ci, c in enumerate(spair):
            load[owners[ci]] += len(c.thresholds)
        eval_share = max(load) / max(total_pairs, 1)
        wall = 0.0
        for name, v in per.items():
            st = by_name.get(name)
            if st is None or "c_s" not in v:
                continue
            vit = st.batch > 32
            c1, cn = launch_cost(c_launch_plan(spair, st.batch, 1), vit), launch_cost(c_launch_plan(spair, st.batch, n), vit)
            a1 = launch_cost(plan_launches(n_a_images, st.batch, equal=False), vit)
            an = launch_cost(plan_launches(-(-n_a_images // n), st.batch, equal=False), vit)
            wall += v.get("c_tower_s", v["c_s"]) * cn / c1 + v.get("c_eval_s", 0.0) * eval_share + v.get("a_s", 0.0) * an / a1 + PER_SETTING_SYNC_S
        out[f"n{n}"] = {"wall_s": round(wall, 2), "speedup": round(base / wall, 2) if wall else None}
    out["from_wall_s"] = round(base, 3)
    out["model"] = ("per setting: c_tower_s x launch-plan cost ratio (ViT launch = images + 24, diffusion launch = images) + c_eval_s x largest owner's pair share "
                    "+ a_s x A-launch cost ratio + 0.025 s of fences; a prediction, not a measurement")
    out["not_modelled"] = ("RCCL all_to_all kernels need CUs while the next tower launch's persistent GEMMs own all of them (VISREP_RESERVE_CUS reserves some "
                           "when world > 1), host launch overhead of graph replays, per-GPU clock spread, rank skew at the two fences per leg")
    return out


# ------------------------------------------------------------------------------------------------ the sweep
def run_sweep(settings: Sequence[Setting] = SETTINGS, n_a_images: int = 100, spair: Optional[Sequence[SpairCategory]] = None,
              device="cuda", build: Optional[Callable[[Setting], object]] = None, pixels: Optional[Callable] = None, a_hooks=None,
              hidden: int = 4096, precision: str = "reference", do_a: bool = True, do_c: bool = True, verbose: bool = False,
              also_bf16: bool = False, fp32_products: Optional[int] = None, alt_fp32_products: Optional[int] = None) -> dict:
    """Runs the sweep on this process' share (one process per GPU; world size from torch.distributed).  Returns
    {"wall_s", "setup_s", "per_setting": {name: {"a_s", "c_s", "A", "pck": [..3], "images", "dtype": {"a", "c"}}}, "images", ...}.
    precision: see SettingModel ('reference' = A leg bf16, C leg in the dtype the reference's C path uses per tower).
    fp32_products: product set of the fp32 ViT engines (SettingModel): None = fp32-equivalent (6); bench.py's throughput sweep opts into 3
    and the per-setting "dtype" / the top-level "fp32_products" say so.
    alt_fp32_products ('reference' only): the settings whose C leg is fp32 run it once more on engines with this product set ("c_s_fp32x<n>",
    "pck_fp32x<n>"; "wall_s_fp32x<n>" = the sweep's wall-clock with those legs swapped in).  bench.py: reference leg = 6 (fp32-equivalent), alt = 3.
    also_bf16 ('reference' only): the settings whose C leg is fp32 run their C leg a second time on the bf16 engines, timed the same way
    ("c_s_bf16", "pck_bf16"), and "wall_s_all_bf16" is the wall-clock of the sweep with those legs swapped in - the all-bf16 sweep's
    number without running the eleven other legs twice (they are the same launches in both modes).
    build / pixels / a_hooks: injection points for the CPU tests (stand-in towers; the oracle as the score kernels)."""
    d = _dist()
    rank, world = (d.get_rank(), d.get_world_size()) if d else (0, 1)
    dev = torch.device(device)
    import logging
    from .C_score import pck_train as _PT  # noqa: F401  (its import configures the logger: do it before silencing it)
    clog = logging.getLogger("visrep.cscore")
    old_level = clog.level
    clog.setLevel(logging.WARNING)                                       # 18 per-category lines x 13 settings are not a bench output
    try:
        return _run_sweep(settings, n_a_images, spair, dev, build, pixels, a_hooks, hidden, precision, do_a, do_c, verbose, rank, world, also_bf16, fp32_products,
                          alt_fp32_products)
    finally:
        clog.setLevel(old_level)


def _run_sweep(settings, n_a_images, spair, dev, build, pixels, a_hooks, hidden, precision, do_a, do_c, verbose, rank, world, also_bf16=False,
               fp32_products=None, alt_fp32_products=None):
    if precision != "reference":
        alt_fp32_products = None
    build = build or (lambda s: SettingModel(s, dev, hidden=hidden, precision=precision, fp32_products=fp32_products, alt_fp32_products=alt_fp32_products))
    # fp32 pixels wherever an fp32 engine may consume them (every tower is fed its own dtype: SettingModel._run); the bf16 cast of the same
    # draw is what the all-bf16 mode generates directly, so the three modes see the same images
    pixels = pixels or ResidentPixels(dev, torch.bfloat16 if precision == "bf16" else torch.float32)
    if spair is None and do_c:
        spair = synthetic_spair()
    per, refs, pending = {}, {}, []
    _, scales_fn = a_hooks or (_a_hooks() if do_a else (None, None))
    wall = setup = wall_swap = wall_alt = 0.0
    alt_key = f"fp32x{int(alt_fp32_products)}" if alt_fp32_products else None
    n_c_images = sum(c.n_images for c in spair) if do_c else 0
    my_a = list(range(rank, n_a_images, world))
    for st in settings:
        t0 = time.perf_counter()
        model = build(st)
        if hasattr(model, "warm"):
            model.warm(launch_shapes(st, n_a_images, spair, rank, world, do_a, do_c))
        if hasattr(pixels, "prefetch"):                                 # inputs resident in HBM before the timed legs start
            pixels.prefetch((my_a if do_a else []) + (c_item_ids(spair, rank, world) if do_c else []), st.size)
        _fence(dev)
        t_setup = time.perf_counter() - t0
        setup += t_setup
        ent = {"setup_s": round(t_setup, 3)}
        if do_a:
            _fence(dev)
            t0 = time.perf_counter()
            feat = a_features(model, n_a_images, my_a, pixels)
            if st.key in REFS:
                refs[st.key] = (feat, scales_fn(feat) if feat is not None else None)
            pending.append((st, feat))
            if all(r in refs for r in REFS):                           # both references exist: score everything that waited for them
                for pst, pf in pending:
                    per.setdefault(pst.name, {})["A"] = a_score_of(pf, refs, dev, a_hooks)
                pending = []
            _fence(dev)
            ent["a_s"] = round(time.perf_counter() - t0, 4)
            wall += time.perf_counter() - t0
        if do_c:
            _fence(dev)
            t0 = time.perf_counter()
            split = {}
            pck = c_score_of(model, spair, pixels, dev, rank, world, timing=split)
            _fence(dev)
            c_s = time.perf_counter() - t0
            ent["c_s"] = round(c_s, 4)
            ent["c_tower_s"], ent["c_eval_s"] = round(split.get("tower_s", 0.0), 4), round(split.get("eval_s", 0.0), 4)
            ent["pck"] = [float(x) for x in pck]
            wall += c_s
            dts = getattr(model, "dtypes", None)
            if alt_key and getattr(model, "alt_c_towers", None) is not None:
                _fence(dev)
                t1 = time.perf_counter()
                pck_x = c_score_of(model, spair, pixels, dev, rank, world, tokens=model.c_tokens_alt)
                _fence(dev)
                ent[f"c_s_{alt_key}"] = round(time.perf_counter() - t1, 4)
                ent[f"pck_{alt_key}"] = [float(x) for x in pck_x]
                wall_alt += (time.perf_counter() - t1) - c_s
            if also_bf16 and precision == "reference" and dts and dts["c"] != dts["a"]:
                _fence(dev)
                t1 = time.perf_counter()
                pck_b = c_score_of(model, spair, pixels, dev, rank, world, tokens=model.tokens)    # the A leg's (bf16) engines on the C images
                _fence(dev)
                ent["c_s_bf16"] = round(time.perf_counter() - t1, 4)
                ent["pck_bf16"] = [float(x) for x in pck_b]
                wall_swap += (time.perf_counter() - t1) - c_s
        if getattr(model, "dtypes", None):
            ent["dtype"] = dict(model.dtypes)
        ent["images"] = (n_a_images if do_a else 0) + n_c_images
        per.setdefault(st.name, {}).update(ent)
        if verbose and rank == 0:
            print(f"[sweep] {st.name}: {per[st.name]}", flush=True)
        del model
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    if pending:
        raise ValueError("the A score needs the clip336 and clip224 settings in the sweep (A_score/compute.py:31-35)")
    images = sum(v["images"] for v in per.values())
    extra = {"wall_s_all_bf16": round(wall + wall_swap, 3)} if also_bf16 and precision == "reference" and do_c else {}
    if alt_key and do_c:
        extra[f"wall_s_{alt_key}"] = round(wall + wall_alt, 3)
    # every rank's setup time (engine construction + HIP-graph capture + resident pixels): outside the timed legs, inside anyone's wall-clock
    setup_ranks = [round(setup, 3)]
    d = _dist()
    if d is not None and world > 1:
        allr = [None] * world
        d.all_gather_object(allr, round(setup, 3))
        setup_ranks = [float(x) for x in allr]
    a_total = sum(v.get("a_s", 0.0) for v in per.values())
    c_total = sum(v.get("c_s", 0.0) for v in per.values())
    # one table-ready row per run: `bench.py --gpus N` at N = 1, 2, 4, 8 gives the north star's scaling line (images/s of the whole sweep,
    # per-leg seconds) without post-processing - the driver computes the efficiency from the rows
    scaling_row = {"n_gpus": world, "wall_s": round(wall, 3), "img_s": round(images / wall, 2) if wall else None,
                   "a_leg_s": round(a_total, 3), "c_leg_s": round(c_total, 3),
                   "c_leg_s_by_setting": {k: v.get("c_s") for k, v in per.items() if "c_s" in v},
                   "setup_s_max_over_ranks": max(setup_ranks), "setup_s_per_rank": setup_ranks, "images": images}
    if world == 1 and do_a and do_c:
        scaling_row["predicted"] = predict_scaling(per, settings, n_a_images, spair)
    return {"wall_s": round(wall, 3), **extra, "setup_s": round(setup, 3), "world": world, "settings": len(per), "images": images,
            "img_s": round(images / wall, 2) if wall else None, "img_s_per_gpu": round(images / wall / world, 2) if wall else None,
            "a_images_per_setting": n_a_images if do_a else 0, "c_images_per_setting": n_c_images,
            "c_pairs_per_setting": sum(len(c.thresholds) for c in spair) if do_c else 0, "tower_precision": precision,
            "fp32_products": ("engine default (6: fp32-equivalent)" if fp32_products is None else
                              f"{int(fp32_products)} (explicit opt-in; 6 = fp32-equivalent, 3 = two-plane operands, mid x mid dropped)"),
            "scaling": "strong (fixed total work, images sharded rank::world, C categories owned by ranks)",
            "pixels": "resident in HBM (drawn in each setting's setup)" if hasattr(pixels, "prefetch") else "drawn by the caller's image source inside the legs",
            "scaling_row": scaling_row, "per_setting": per}


# ------------------------------------------------------------------------------------------------ encoder-sharded A score (configs[2])
def a_scores_encoder_sharded(settings: Sequence[Setting], n_images: int, device="cuda", build=None, pixels=None, a_hooks=None, chunk: int = 32,
                             hidden: int = 4096, precision: str = "bf16") -> Dict[str, float]:     # the A leg is bf16 in 'reference' too
    """BASELINE.json configs[2]: "A_score ... encoder-sharded".  The two CLIP reference stacks are needed by every encoder, so they
    are produced image-sharded (every rank runs CLIP336 / CLIP224 on images i = rank mod world) and ALL-GATHERED in chunks of `chunk`
    images per rank: the gather of chunk k is issued asynchronously (it runs on the collective's own stream) and chunk k + 1's tower
    forward is enqueued behind it on the compute stream, so the xGMI transfer hides under the towers.  Every other setting is owned
    by ONE rank (setting j of the non-reference list on rank j mod world), which runs its tower + projector over ALL images and
    scores them against the gathered references; the per-encoder results are exchanged with one all_gather_object at the end.
    Returns {setting name: A score} on every rank - equal to the single-process / image-sharded numbers."""
    d = _dist()
    rank, world = (d.get_rank(), d.get_world_size()) if d else (0, 1)
    dev = torch.device(device)
    build = build or (lambda s: SettingModel(s, dev, hidden=hidden, precision=precision))
    pixels = pixels or (lambda ids, size: synthetic_pixels(ids, size, dev))
    score, scales = a_hooks or _a_hooks()
    by_key = {s.key: s for s in settings}
    if any(r not in by_key for r in REFS):
        raise ValueError("the A score needs the clip336 and clip224 settings (A_score/compute.py:31-35)")
    refs = {}
    per = (n_images + world - 1) // world
    for r in REFS:
        model = build(by_key[r])
        mine = list(range(rank, n_images, world))
        pieces, inflight = [], None
        for s in range(0, per, chunk):
            ids = mine[s:s + chunk]
            n_here = min(chunk, per - s)                              # rows this chunk holds per rank (padded on short ranks)
            f = model.project(model.tokens(pixels(ids, model.setting.size))) if ids else None
            if f is None:
                probe = model.project(model.tokens(pixels([0], model.setting.size)))
                f = probe[:0]
            if inflight is not None:
                pieces.append(inflight())                              # the previous chunk's gather had this chunk's forward to hide under
            n_glob = min(n_images - s * world, n_here * world)
            _, inflight = all_gather_rows(f.contiguous(), n_glob, rank, world, async_op=True)
        if inflight is not None:
            pieces.append(inflight())
        full = torch.cat(pieces, 0)                                   # chunk c holds global rows [c * chunk * world, ...), in order
        refs[r] = (full, scales(full))
        del model
    results = {}
    others = [s for s in settings]
    for j, st in enumerate(others):
        if j % world != rank:
            continue
        if st.key in REFS:
            feat = refs[st.key][0]
        else:
            model = build(st)
            feat = a_features(model, n_images, list(range(n_images)), pixels)
            del model
        fs = scales(feat)
        tot = []
        for r in REFS:
            ref, rs = refs[r]
            s = score(feat, ref, fs, rs).double().cpu()
            tot.append(sum(float(s[i]) for i in range(n_images)) / n_images)
        results[st.name] = (tot[0] + tot[1]) / 2
    if d is not None:
        allr = [None] * world
        d.all_gather_object(allr, results)
        results = {k: v for part in allr for k, v in part.items()}
    return {s.name: results[s.name] for s in settings}


def main(argv=None):
    import argparse
    import json
    ap = argparse.ArgumentParser(description="13-setting A + C sweep (MI355X)")
    ap.add_argument("--a-images", type=int, default=100)
    ap.add_argument("--c-images", type=int, default=1800)
    ap.add_argument("--c-pairs", type=int, default=12234)
    ap.add_argument("--settings", nargs="*", default=None, help="subset of setting names (default: all 13)")
    ap.add_argument("--precision", default="reference", choices=["reference", "bf16", "fp32"],
                    help="reference = A leg bf16, C leg in the reference's per-tower dtype (fp32 for CLIP / OpenCLIP / DINOv2); bf16 / fp32 = both legs")
    ap.add_argument("--also-bf16", action="store_true", help="reference mode: time the fp32 C legs on the bf16 engines too (wall_s_all_bf16)")
    ap.add_argument("--mode", default="image", choices=["image", "encoder"], help="image-sharded sweep, or the encoder-sharded A score only")
    ap.add_argument("--fp32-products", type=int, default=None, choices=[3, 4, 6],
                    help="split-bf16 product set of the fp32 ViT engines (default: the engine's fp32-equivalent 6; 3 = throughput opt-in)")
    a = ap.parse_args(argv)
    from . import dist_env
    owned = dist_env.init_from_env()
    try:
        sel = [s for s in SETTINGS if a.settings is None or s.name in a.settings]
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        if a.mode == "encoder":
            out = a_scores_encoder_sharded(sel, a.a_images, dev, precision="fp32" if a.precision == "fp32" else "bf16")
        else:
            out = run_sweep(sel, a.a_images, synthetic_spair(a.c_images, a.c_pairs), dev, precision=a.precision, verbose=True, also_bf16=a.also_bf16,
                            fp32_products=a.fp32_products)
        if not _dist() or _dist().get_rank() == 0:
            print(json.dumps(out), flush=True)
        return out
    finally:
        dist_env.finalize(owned)


if __name__ == "__main__":
    main()
