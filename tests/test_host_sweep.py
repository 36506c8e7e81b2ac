"""Host logic of the 13-setting A + C sweep driver (law_of_vision_representation_in_mllms_amd/sweep.py) on CPU: the sharding, the
all-gather of the C-score feature bank, the A-score all-reduce and the encoder-sharded A score (BASELINE.json configs[2] / [4]),
single process and on 2 gloo ranks.  Towers are deterministic CPU stand-ins, the score kernels are the oracle (monkeypatched
hooks, like tests/test_host_cscore.py / test_host_ascore.py) - the product has no CPU fallback; this exercises the driver only."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(__file__))
from test_host_cscore import cpu_pck_counts, cpu_transfer  # noqa: E402

from law_of_vision_representation_in_mllms_amd import cscore_ops, sweep as S  # noqa: E402
from oracle import ascore as OA, cscore as OC  # noqa: E402

D_OUT = 32
SETTINGS = (S.Setting("CLIP336", "clip336", ("a",), 24, 3), S.Setting("CLIP224", "clip224", ("b",), 16, 4),
            S.Setting("DINOv2", "dino", ("c",), 16, 3), S.Setting("CLIP224+DINOv2", "clip224+dino", ("b", "c"), 16, 2))


class FakeModel:
    """tokens = 4x4 mean-pooled pixels through a fixed linear map per tower id; project = another fixed linear map."""

    def __init__(self, st):
        self.setting = st
        self.maps = []
        for tid in st.towers:
            g = torch.Generator().manual_seed(sum(map(ord, tid)))
            self.maps.append(torch.randn(3 * 16, 8, generator=g))
        g = torch.Generator().manual_seed(99 + len(st.towers))
        self.proj = torch.randn(8 * len(st.towers), D_OUT, generator=g)
        self.split = 8 if len(st.towers) == 2 else 0

    def tokens(self, px):
        B, _, s, _ = px.shape
        p = px.float().unfold(2, 4, 4).unfold(3, 4, 4)                      # [B, 3, s/4, s/4, 4, 4]
        p = p.permute(0, 2, 3, 1, 4, 5).reshape(B, (s // 4) ** 2, 48)
        return torch.cat([p @ m for m in self.maps], -1)

    def project(self, tok):
        return tok @ self.proj


def fake_pixels(ids, size):
    out = torch.empty(len(ids), 3, size, size)
    g = torch.Generator()
    for j, i in enumerate(ids):
        g.manual_seed(1000 + int(i))
        out[j] = torch.rand(3, size, size, generator=g) * 2 - 1
    return out


def oracle_score(o, r, o_scale=None, r_scale=None):
    return torch.tensor([OA.max_cos_mean(o[i], r[i]) for i in range(o.shape[0])])


HOOKS = (oracle_score, lambda x: None)


def spair_small():
    return S.synthetic_spair(n_images=14, n_pairs=20, kmax=6, seed=3, categories=("aeroplane", "cat", "dog"))


def run(world_rank=None):
    return S.run_sweep(SETTINGS, n_a_images=7, spair=spair_small(), device="cpu", build=FakeModel, pixels=fake_pixels, a_hooks=HOOKS)


@pytest.fixture
def cpu_ops(monkeypatch):
    monkeypatch.setattr(cscore_ops, "transfer", cpu_transfer)
    monkeypatch.setattr(cscore_ops, "pck_counts", cpu_pck_counts)


def direct_a(name):
    st = {s.name: s for s in SETTINGS}
    f = lambda s: list(FakeModel(s).project(FakeModel(s).tokens(fake_pixels(range(7), s.size))))
    return OA.a_score(f(st[name]), f(st["CLIP336"]), f(st["CLIP224"]))[0]


def direct_c(st):
    model = FakeModel(st)
    per_cat, weights = [], []
    for ci, cat in enumerate(spair_small()):
        maps = model.tokens(fake_pixels([ci * 100000 + i for i in range(cat.n_images)], st.size))
        P = int(round(maps.shape[1] ** 0.5))
        feats = []
        for s in cat.slot:
            m = maps[int(s)]
            if model.split:
                m = OC.normalize_feats_two(m[None], model.split)[0]
            feats.append(m.t().reshape(1, -1, P, P))
        N = len(cat.thresholds)
        _, img_correct, _ = OC.category_pck(feats, list(range(N)), cat.kps, cat.thresholds, P)
        per_cat.append(img_correct[:3])
        weights.append(N)
    return OC.weighted_pcks(per_cat, weights)


def test_sweep_single_process_equals_the_oracle_chain(cpu_ops):
    out = run()
    assert out["settings"] == 4 and out["world"] == 1
    assert out["images"] == 4 * (7 + 14) and out["c_pairs_per_setting"] == 20
    row = out["scaling_row"]                                                  # the row the driver's N = 1, 2, 4, 8 runs line up
    assert row["n_gpus"] == 1 and row["images"] == out["images"] and len(row["setup_s_per_rank"]) == 1
    assert set(row["c_leg_s_by_setting"]) == {st.name for st in SETTINGS} and abs(row["a_leg_s"] + row["c_leg_s"] - row["wall_s"]) < 1e-2
    assert row["setup_s_max_over_ranks"] == max(row["setup_s_per_rank"]) and row["img_s"] > 0
    for st in SETTINGS:
        ent = out["per_setting"][st.name]
        assert abs(ent["A"] - direct_a(st.name)) < 1e-9, st.name
        np.testing.assert_allclose(ent["pck"], direct_c(st), atol=1e-7, err_msg=st.name)


def test_sweep_needs_both_references(cpu_ops):
    with pytest.raises(ValueError, match="clip336 and clip224"):
        S.run_sweep(SETTINGS[2:3], n_a_images=3, spair=spair_small(), device="cpu", build=FakeModel, pixels=fake_pixels, a_hooks=HOOKS, do_c=False)


def test_all_gather_rows_single_process_is_identity():
    x = torch.arange(12.).view(4, 3)
    assert torch.equal(S.all_gather_rows(x, 4, 0, 1), x)
    h, fin = S.all_gather_rows(x, 4, 0, 1, async_op=True)
    assert h is None and torch.equal(fin(), x)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cscore_ops.transfer, cscore_ops.pck_counts = cpu_transfer, cpu_pck_counts
    out = S.run_sweep(SETTINGS, n_a_images=7, spair=spair_small(), device="cpu", build=FakeModel, pixels=fake_pixels, a_hooks=HOOKS)
    enc = S.a_scores_encoder_sharded(SETTINGS, 7, device="cpu", build=FakeModel, pixels=fake_pixels, a_hooks=HOOKS, chunk=2)
    x = torch.arange(10.)[rank::world].view(-1, 1) * torch.ones(1, 3)
    gathered = S.all_gather_rows(x, 10, rank, world)
    q.put((rank, out, enc, gathered.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sweep_and_encoder_sharded_a_score_equal_single_process(cpu_ops):
    single = run()
    enc_single = S.a_scores_encoder_sharded(SETTINGS, 7, device="cpu", build=FakeModel, pixels=fake_pixels, a_hooks=HOOKS, chunk=2)
    for st in SETTINGS:
        assert abs(enc_single[st.name] - single["per_setting"][st.name]["A"]) < 1e-12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, enc, gathered in got:
        assert out["world"] == 2 and out["images"] == single["images"]
        row = out["scaling_row"]                                              # every rank reports every rank's setup time
        assert row["n_gpus"] == 2 and len(row["setup_s_per_rank"]) == 2 and row["setup_s_max_over_ranks"] == max(row["setup_s_per_rank"])
        np.testing.assert_array_equal(gathered[:, 0], np.arange(10.))          # global row order restored, padding dropped
        for st in SETTINGS:
            assert abs(out["per_setting"][st.name]["A"] - single["per_setting"][st.name]["A"]) < 1e-12, st.name
            np.testing.assert_allclose(out["per_setting"][st.name]["pck"], single["per_setting"][st.name]["pck"], atol=1e-12)
            assert abs(enc[st.name] - single["per_setting"][st.name]["A"]) < 1e-12, st.name


def test_shard_plan_keeps_launches_full_at_world_8():
    """the per-category tower launches were 12-13 images at world 8.  Now one pass over a setting's 1,800 C images:
    every rank's launches stay near the setting's launch batch, all of one or two shapes, and no collective is needed for the pairs."""
    spair = S.synthetic_spair()
    assert sum(c.n_images for c in spair) == 1800 and sum(len(c.thresholds) for c in spair) == 12234
    for world in (1, 2, 4, 8):
        for st in S.SETTINGS:
            plan = S.c_launch_plan(spair, st.batch, world)
            per = -(-1800 // world)
            assert sum(plan) == per and max(plan) <= st.batch and len(set(plan)) <= 2
            if st.batch == 32:
                assert min(plan) >= 28 and max(plan) - min(plan) <= 1    # diffusion towers: 225 images -> 8 launches of 28 / 29
            else:
                assert all(q == st.batch for q in plan[:-1])             # ViT towers: full launches (sized for the GEMMs' tile rounds) + a remainder
            shapes = S.launch_shapes(st, 100, spair, world - 1, world)
            assert set(plan) <= set(shapes) and len(set(shapes)) <= 4     # what SettingModel.warm() captures in setup
        own = S.category_owners(spair, world)
        load = [sum(len(c.thresholds) for c, o in zip(spair, own) if o == r) for r in range(world)]
        assert sorted(set(own)) == list(range(world)) and max(load) <= 12234 / world + max(len(c.thresholds) for c in spair)
    assert S.plan_launches(0, 16) == [] and S.plan_launches(5, 16) == [5] and S.plan_launches(33, 16) == [11, 11, 11]
    assert S.plan_launches(1800, 256, equal=False) == [256] * 7 + [8] and S.plan_launches(512, 256, equal=False) == [256, 256]


def test_settings_table_is_the_papers():
    """policy/fit.py:20 lists the 13 settings; every tower id is in the drop-in registry; the A keys are compute.py:10's names."""
    from law_of_vision_representation_in_mllms_amd.llava.model.llava_arch import build_function_mapping
    assert [s.name for s in S.SETTINGS] == ["CLIP336", "CLIP224", "OpenCLIP", "DINOv2", "SDim", "SD1.5", "SDXL", "DiT", "SD3", "SD2.1", "SigLIP",
                                            "CLIP224+DINOv2", "CLIP336+DINOv2"]
    assert all(t in build_function_mapping for s in S.SETTINGS for t in s.towers)
    assert {"clip336", "clip224", "dino", "dit", "imsd", "openclip", "sd1.5", "sd2.1", "sd3", "sdxl"} <= {s.key for s in S.SETTINGS}
    sp = S.synthetic_spair()
    assert len(sp) == 18 and sum(len(c.thresholds) for c in sp) == 12234 and sum(c.n_images for c in sp) == 1800


class TwoLegModel(FakeModel):
    """A stand-in with the reference mode's two engine sets: the C leg sees `c_tokens` (here: the A-leg tokens, perturbed), `dtypes` says
    the legs differ."""

    def __init__(self, st):
        super().__init__(st)
        self.dtypes = {"a": "bf16", "c": "fp32"} if st.name != "DINOv2" else {"a": "bf16", "c": "bf16"}
        self.calls = {"a": 0, "c": 0}

    def tokens(self, px):
        self.calls["a"] += 1
        return super().tokens(px)

    def c_tokens(self, px):
        self.calls["c"] += 1
        return super().tokens(px) * 1.0 + (0.25 if self.dtypes["c"] == "fp32" else 0.0)


def test_reference_mode_runs_the_c_leg_on_its_own_engines_and_times_the_bf16_twin(cpu_ops):
    """precision='reference': the C leg goes through the model's C-leg engines (c_tokens), the A leg through tokens; with also_bf16 the
    settings whose legs differ run the C leg once more on the A-leg engines ("c_s_bf16", "pck_bf16") and the all-bf16 wall-clock is the
    reference wall-clock with those legs swapped."""
    models = {}

    def build(st):
        models[st.name] = TwoLegModel(st)
        return models[st.name]
    out = S.run_sweep(SETTINGS, n_a_images=7, spair=spair_small(), device="cpu", build=build, pixels=fake_pixels, a_hooks=HOOKS, precision="reference",
                      also_bf16=True)
    plain = run()
    assert out["tower_precision"] == "reference" and "wall_s_all_bf16" in out
    for st in SETTINGS:
        ent, m = out["per_setting"][st.name], models[st.name]
        assert ent["dtype"] == m.dtypes
        assert abs(ent["A"] - plain["per_setting"][st.name]["A"]) < 1e-12                   # the A leg never sees the C-leg engines
        if m.dtypes["c"] != m.dtypes["a"]:
            assert m.calls["c"] > 0 and "c_s_bf16" in ent
            np.testing.assert_allclose(ent["pck_bf16"], plain["per_setting"][st.name]["pck"], atol=1e-12)      # the twin = the A-leg engines
        else:
            assert "c_s_bf16" not in ent
            np.testing.assert_allclose(ent["pck"], plain["per_setting"][st.name]["pck"], atol=1e-12)
    swap = sum(e["c_s_bf16"] - e["c_s"] for e in out["per_setting"].values() if "c_s_bf16" in e)
    assert abs(out["wall_s_all_bf16"] - (out["wall_s"] + swap)) < 0.05
    with pytest.raises(ValueError, match="precision"):
        S.SettingModel(SETTINGS[0], "cpu", precision="fp16")


def test_reference_dtypes_follow_the_reference_scripts():
    """C_score/extract_feature.py:36-50,80-91: CLIP / OpenCLIP / DINOv2 are built without a dtype cast (fp32), SigLIP and the diffusion
    towers in bf16; the A path is LLaVA's model.to(bfloat16)."""
    want = {S.CLIP336: "fp32", S.CLIP224: "fp32", S.OPENCLIP: "fp32", S.DINOV2: "fp32", S.SIGLIP: "bf16", S.SD15: "bf16", S.SD21: "bf16",
            S.SDXL: "bf16", S.IMSD: "bf16", S.DIT: "bf16", S.SD3: "bf16"}
    assert {t: S.reference_c_precision(t) for t in want} == want


def test_c_exchange_plan_sends_every_row_once_to_its_owner():
    """the C leg all-gathered every launch's maps to every rank although only the category owner needs them.  The
    owner-addressed plan: every real row leaves its producer exactly once, for the rank that owns its category; padding rows (a short
    rank's repeated image) go nowhere; summed over a setting the fabric carries n_items rows instead of world * n_items."""
    spair = S.synthetic_spair()
    items = [(ci, i) for ci, cat in enumerate(spair) for i in range(cat.n_images)]
    for world in (1, 2, 8):
        owner = S.category_owners(spair, world)
        item_owner = [owner[ci] for ci, _ in items]
        plan = S.c_launch_plan(spair, 16, world)
        seen, off, rows_on_fabric = set(), 0, 0
        for sz in plan:
            send = S.c_exchange_plan(len(items), item_owner, world, off, sz)
            for r in range(world):
                for q in range(world):
                    assert send[r][q] == sorted(send[r][q])
                    for j in send[r][q]:
                        g = (off + j) * world + r
                        assert 0 <= j < sz and g < len(items) and item_owner[g] == q and g not in seen
                        seen.add(g)
                        rows_on_fabric += int(q != r)
            off += sz
        assert seen == set(range(len(items)))                       # every image's maps reach their owner, once
        assert rows_on_fabric <= len(items) and (world == 1) == (rows_on_fabric == 0)


def test_reference_precision_chunks_fill_whole_tile_rounds():
    """engine.best_chunk prices a launch with the GEMM dispatcher's own rule (csrc/gemm_bf16.hip: full rounds of 256 x 256 tiles on the
    device's CUs; a remainder of rem tiles goes to the 128 x 128 tail launch when 4 rem <= CUs, else it costs a whole round) and returns the
    largest image count within 1 % of the best efficiency.  Pinned values for MI355X (256 CUs): the numbers the comments in sweep.py,
    bench.py and VitEngineF32.chunk quote."""
    from law_of_vision_representation_in_mllms_amd.engine import best_chunk

    def rounds(c, T, d=1024):
        tiles = -(-c * T // 256) * (d // 256)
        return divmod(tiles, 256)
    assert best_chunk(577, 1024, 128, cus=256) == 113 and rounds(113, 577) == (3, 252)        # 3.98 rounds, 4 run; 64 images (the old chunk) = 2.25: 3 run
    assert best_chunk(257, 1024, 128, cus=256) == 127 and rounds(127, 257) == (2, 0)          # exactly two rounds, no tail launch (128 images: + a 4-tile tail pair)
    assert best_chunk(257, 1024, 256, cus=256) == 255 and rounds(255, 257) == (4, 0)
    assert best_chunk(577, 1024, 1, cus=256) == 1 and best_chunk(50, 768, 7, cus=256) == 7   # tiny caps: whatever fits
    assert best_chunk(577, 1024, 128) == 113                                                  # no device visible here: the MI355X count
    # another part (304 CUs): the choice follows the CU count instead of silently de-tuning
    c = best_chunk(577, 1024, 128, cus=304)
    tiles = -(-c * 577 // 256) * 4
    assert tiles % 304 == 0 or tiles % 304 > 304 * 0.9 or (tiles % 304) * 4 <= 304


def test_resident_pixels_are_the_same_images():
    """sweep.ResidentPixels (the default image source: drawn in setup, index_select in the timed legs) returns exactly synthetic_pixels' images,
    whatever was prefetched, in whatever order and batch shape; c_item_ids lists the ids c_score_of asks for."""
    src = S.ResidentPixels("cpu", torch.float32)
    src.prefetch([5, 100003, 7, 5], 12)
    src.prefetch([9, 7], 12)                                                 # grows the store, keeps the rows
    src.prefetch([1, 2], 20)
    for ids, size in (([7, 5, 9], 12), ([100003], 12), ([2, 1, 2], 20), ([5, 6], 12), ([3], 16)):       # the last two: not (all) prefetched
        assert torch.equal(src(ids, size), S.synthetic_pixels(ids, size, "cpu", torch.float32))
    spair = S.synthetic_spair(60, 90)
    for world in (1, 3):
        got = [S.c_item_ids(spair, r, world) for r in range(world)]
        n = sum(c.n_images for c in spair)
        assert all(len(g) == -(-n // world) for g in got)
        flat = sorted({i for g in got for i in g})
        assert flat == sorted(ci * 100000 + i for ci, c in enumerate(spair) for i in range(c.n_images))


# ------------------------------------------------------------------------------------------------ world 8 as eight real processes
TWIN_SETTINGS = (S.Setting("CLIP336", "clip336", ("clip_a",), 42, 5), S.Setting("CLIP224", "clip224", ("clip_b",), 28, 7),
                 S.Setting("DINOv2", "dino", ("dino_c",), 28, 4), S.Setting("CLIP224+DINOv2", "clip224+dino", ("clip_b", "dino_c"), 28, 3))


class TwinModel:
    """Tiny towers on the product's HOST twins (engine.VitEngineCPU = visrep_vit_forward_cpu; no oracle, no torch model): what a rank of the
    sweep runs when its device is "cpu".  Projector = one fixed linear map (the sweep driver is what is under test)."""

    def __init__(self, st):
        from law_of_vision_representation_in_mllms_amd import engine, vit_weights as VW
        self.setting = st
        self.engines = []
        for tid in st.towers:
            fam = "clip" if tid.startswith("clip") else "dinov2"
            spec = VW.tiny_spec(fam, image_size=st.size, patch=7, d=32, heads=2, mlp=64, layers=2)
            w = VW.synthetic_weights(spec, seed=sum(map(ord, tid)))
            self.engines.append((spec, engine.VitEngineCPU(spec, w, threads=1)))
        g = torch.Generator().manual_seed(5 + len(st.towers))
        self.proj = torch.randn(32 * len(st.towers), D_OUT, generator=g) * 0.2
        self.split = 32 if len(st.towers) == 2 else 0

    def tokens(self, px):
        return torch.cat([e.forward(px.float(), n_layers=1)[:, (1 if s.has_cls else 0):] for s, e in self.engines], -1)

    def project(self, tok):
        return tok @ self.proj


def _twin_ops():
    """the score kernels' host twins behind the device entry points' signatures (explicit, as a device="cpu" caller wires them)"""
    from law_of_vision_representation_in_mllms_amd import ascore_ops
    cscore_ops.transfer = lambda bank, *a, **k: cscore_ops.transfer_cpu(bank.float().contiguous(), *a, **{x: v for x, v in k.items() if x not in ("packed", "sort_pairs")})
    cscore_ops.pck_counts = cscore_ops.pck_counts_cpu
    return (lambda o, r, os_=None, rs_=None: ascore_ops.max_cos_mean_cpu(o, r, threads=1)), (lambda x: None)


def _twin_run():
    hooks = _twin_ops()
    return S.run_sweep(TWIN_SETTINGS, n_a_images=19, spair=S.synthetic_spair(n_images=21, n_pairs=30, kmax=6, seed=4, categories=("aeroplane", "cat", "dog", "bus", "tvmonitor")),
                       device="cpu", build=TwinModel, pixels=fake_pixels, a_hooks=hooks)


def _twin_worker(rank, world, port, q):
    import torch.distributed as dist
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _twin_run()
    q.put((rank, {k: (v["A"], v["pck"]) for k, v in out["per_setting"].items()}, out["world"], out["images"], out["scaling_row"]["n_gpus"]))
    dist.barrier()
    dist.destroy_process_group()


def test_world_8_processes(monkeypatch):
    """The WHOLE sweep (A leg with its all-reduce, C leg with the owner-addressed all_to_all exchange and the all_gather_object of the category
    results) as EIGHT gloo processes on the host twins - 19 A images and 21 C images over 8 ranks: ragged shares, ranks whose share of a
    launch is empty, owners without a category - equals the single-process run: A to 1e-6 relative (fp32 sums re-associated by the all-reduce),
    PCK exactly."""
    monkeypatch.setattr(cscore_ops, "transfer", cscore_ops.transfer)          # _twin_ops rebinds them: restore after the test
    monkeypatch.setattr(cscore_ops, "pck_counts", cscore_ops.pck_counts)
    single = _twin_run()
    assert single["world"] == 1 and "predicted" in single["scaling_row"]
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34500 + os.getpid() % 2000
    procs = [ctx.Process(target=_twin_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(g[0] for g in got) == list(range(world))
    for rank, per, w, images, n in got:
        assert w == world and n == world and images == single["images"]
        for st in TWIN_SETTINGS:
            a1, pck1 = single["per_setting"][st.name]["A"], single["per_setting"][st.name]["pck"]
            assert abs(per[st.name][0] - a1) <= 1e-6 * abs(a1), (rank, st.name, per[st.name][0], a1)
            np.testing.assert_array_equal(per[st.name][1], pck1, err_msg=f"rank {rank} {st.name}")


# ------------------------------------------------------------------------------------------------ round 6: the x3 leg beside the x6 one; the scaling prediction
class ThreeLegModel(TwoLegModel):
    """reference mode with a second fp32 product set: `alt_c_towers` / `c_tokens_alt` (SettingModel(alt_fp32_products=...))"""

    def __init__(self, st):
        super().__init__(st)
        self.alt_c_towers = [object()] if self.dtypes["c"] == "fp32" else None
        if self.alt_c_towers:
            self.dtypes["c_alt"] = "fp32[split-bf16 x3]"
        self.calls["alt"] = 0

    def c_tokens_alt(self, px):
        self.calls["alt"] += 1
        return FakeModel.tokens(self, px) + 0.25                      # the same maps as the C leg: the alternative set changes speed, not results


def test_alt_product_set_runs_beside_the_reference_leg_and_the_split_timing_adds_up(cpu_ops):
    """bench.py's sweep since round 6: `wall_s` = the fp32-equivalent (x6) C legs, `wall_s_fp32x3` = the same sweep with those legs swapped for the
    throughput set's (timed the same way), per-setting "c_s_fp32x3" / "pck_fp32x3"; the C leg's wall-clock is split into its tower part and its
    evaluation part (what predict_scaling shards differently)."""
    models = {}

    def build(st):
        models[st.name] = ThreeLegModel(st)
        return models[st.name]
    out = S.run_sweep(SETTINGS, n_a_images=7, spair=spair_small(), device="cpu", build=build, pixels=fake_pixels, a_hooks=HOOKS, precision="reference",
                      also_bf16=True, fp32_products=6, alt_fp32_products=3)
    assert "wall_s_fp32x3" in out and "wall_s_all_bf16" in out
    swap = 0.0
    for st in SETTINGS:
        ent, m = out["per_setting"][st.name], models[st.name]
        assert abs(ent["c_tower_s"] + ent["c_eval_s"] - ent["c_s"]) < 5e-3 and ent["c_tower_s"] > 0 and ent["c_eval_s"] > 0
        if m.alt_c_towers:
            assert m.calls["alt"] > 0 and ent["dtype"]["c_alt"] == "fp32[split-bf16 x3]"
            np.testing.assert_allclose(ent["pck_fp32x3"], ent["pck"], atol=1e-12)
            swap += ent["c_s_fp32x3"] - ent["c_s"]
        else:
            assert m.calls["alt"] == 0 and "c_s_fp32x3" not in ent
    assert abs(out["wall_s_fp32x3"] - (out["wall_s"] + swap)) < 0.05
    # outside the reference mode the alternative set is ignored
    plain = S.run_sweep(SETTINGS, n_a_images=7, spair=spair_small(), device="cpu", build=ThreeLegModel, pixels=fake_pixels, a_hooks=HOOKS, precision="bf16",
                        alt_fp32_products=3)
    assert "wall_s_fp32x3" not in plain


def test_predict_scaling_is_arithmetic_on_the_measured_legs():
    """sweep.predict_scaling: per setting c_tower_s x (a rank's launch-plan cost at world N / at world 1) + c_eval_s x the largest owner's share of
    the pairs + a_s x the A-launch cost ratio + 25 ms; ViT launches cost images + 24 image-times, diffusion launches their images.  A model printed
    beside the measured row, never a measurement - the test pins the arithmetic and the fields the bench line carries."""
    spair = S.synthetic_spair()
    settings = (S.Setting("V", "v", ("x",), 336, 256), S.Setting("D", "d", ("y",), 768, 16))
    per = {"V": {"a_s": 0.04, "c_s": 3.5, "c_tower_s": 3.3, "c_eval_s": 0.2}, "D": {"a_s": 0.4, "c_s": 7.7, "c_tower_s": 7.5, "c_eval_s": 0.2}}
    out = S.predict_scaling(per, settings, 100, spair)
    assert set(out) >= {"n2", "n4", "n8", "from_wall_s", "model", "not_modelled"} and abs(out["from_wall_s"] - 11.64) < 1e-9
    assert "RCCL" in out["not_modelled"] and "prediction" in out["model"]
    n_pairs = sum(len(c.thresholds) for c in spair)
    for n in (2, 4, 8):
        owners = S.category_owners(spair, n)
        load = [0] * n
        for ci, c in enumerate(spair):
            load[owners[ci]] += len(c.thresholds)
        want = 0.0
        for st in settings:
            vit = st.batch > 32
            cost = lambda plan: sum(x + (24 if vit else 0) for x in plan)
            v = per[st.name]
            want += v["c_tower_s"] * cost(S.c_launch_plan(spair, st.batch, n)) / cost(S.c_launch_plan(spair, st.batch, 1))
            want += v["c_eval_s"] * max(load) / n_pairs
            want += v["a_s"] * cost(S.plan_launches(-(-100 // n), st.batch, equal=False)) / cost(S.plan_launches(100, st.batch, equal=False)) + 0.025
        assert abs(out[f"n{n}"]["wall_s"] - round(want, 2)) < 1e-9, (n, out[f"n{n}"], want)
        assert 1.0 < out[f"n{n}"]["speedup"] <= n
    assert out["n8"]["wall_s"] < out["n4"]["wall_s"] < out["n2"]["wall_s"] < out["from_wall_s"]
