"""Host logic of the A-score drop-in (file layout, skipping rules, aggregation, rank sharding + all-reduce) on CPU.

The device kernel is replaced by the CPU oracle through the module's `_score_batch` hook — the oracle is the checker
here, the product path itself has no CPU fallback (see test_abi_symbols.py::test_compute_paths_fail_loudly_without_gpu).
"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from law_of_vision_representation_in_mllms_amd.A_score import compute as AC
from oracle import ascore as OA


def _oracle_batch(o, r, o_scale=None, r_scale=None):
    assert o_scale is None and r_scale is None                      # the stand-in below hands out no precomputed factors
    return torch.tensor([OA.max_cos_mean(o[i], r[i]) for i in range(o.shape[0])])


def _no_scales(x):
    return None


def _make_tree(tmp, n=6, D=32):
    rs = np.random.RandomState(0)
    toks = dict(clip336=9, clip224=5, encA=7, encB=4)
    data = {}
    for sub, nt in toks.items():
        os.makedirs(f"{tmp}/{sub}")
        data[sub] = []
        for i in range(1, n + 1):
            t = torch.from_numpy(rs.standard_normal((nt, D)).astype(np.float32))
            torch.save(t, f"{tmp}/{sub}/tensor_{i}.pt")
            data[sub].append(t)
    os.makedirs(f"{tmp}/broken")
    torch.save(torch.zeros(3, D), f"{tmp}/broken/tensor_1.pt")          # only 1 of n files -> encoder skipped
    return data


def test_compute_matches_oracle_and_skips_incomplete_encoders(tmp_path, monkeypatch, capsys):
    data = _make_tree(str(tmp_path))
    monkeypatch.setattr(AC, "_score_batch", _oracle_batch)
    monkeypatch.setattr(AC, "_row_scales", _no_scales)
    res = AC.compute(str(tmp_path), ["clip336", "encA", "broken", "encB"], n_images=6, device="cpu")
    out = capsys.readouterr().out
    assert "Skipping broken due to loading error." in out
    assert set(res) == {"clip336", "encA", "encB"}
    for enc in res:
        want, _, _ = OA.a_score(data[enc], data["clip336"], data["clip224"])
        assert abs(res[enc] - want) < 1e-12
        assert f"Average cosine similarity between clip224+clip336 and {enc}: {res[enc]}" in out


def test_missing_reference_raises(tmp_path, monkeypatch):
    os.makedirs(tmp_path / "encA")
    monkeypatch.setattr(AC, "_score_batch", _oracle_batch)
    monkeypatch.setattr(AC, "_row_scales", _no_scales)
    with pytest.raises(ValueError, match="clip336"):
        AC.compute(str(tmp_path), ["encA"], n_images=2, device="cpu")


def _worker(rank, world, tmp, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    AC._score_batch, AC._row_scales = _oracle_batch, _no_scales
    res = AC.compute(tmp, ["encA", "encB"], n_images=6, device="cpu", verbose=False)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(tmp_path, monkeypatch):
    data = _make_tree(str(tmp_path))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, str(tmp_path), port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for enc in ("encA", "encB"):
        want, _, _ = OA.a_score(data[enc], data["clip336"], data["clip224"])
        assert abs(got[0][enc] - want) < 1e-9 and abs(got[1][enc] - want) < 1e-9


def test_token_stacks_are_normalised_once(tmp_path, monkeypatch):
    """The row factors of an encoder's tokens serve both references; a reference stack and its factors serve every encoder."""
    data = _make_tree(str(tmp_path))
    made, used = [], []

    def scales(x):
        made.append(tuple(x.shape))
        return torch.full(x.shape[:2], float(len(made)))

    def score(o, r, o_scale=None, r_scale=None):
        used.append((tuple(o.shape), tuple(r.shape), float(o_scale[0, 0]), float(r_scale[0, 0])))
        return torch.tensor([OA.max_cos_mean(o[i], r[i]) for i in range(o.shape[0])])
    monkeypatch.setattr(AC, "_score_batch", score)
    monkeypatch.setattr(AC, "_row_scales", scales)
    res = AC.compute(str(tmp_path), ["encA", "encB"], n_images=6, device="cpu", verbose=False)
    assert made == [(6, 7, 32), (6, 9, 32), (6, 5, 32), (6, 4, 32)]            # encA, clip336, clip224, encB - nothing twice
    assert used == [((6, 7, 32), (6, 9, 32), 1.0, 2.0), ((6, 7, 32), (6, 5, 32), 1.0, 3.0),
                    ((6, 4, 32), (6, 9, 32), 4.0, 2.0), ((6, 4, 32), (6, 5, 32), 4.0, 3.0)]
    for enc in res:
        assert abs(res[enc] - OA.a_score(data[enc], data["clip336"], data["clip224"])[0]) < 1e-12
