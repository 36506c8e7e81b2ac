"""The command-line entry points under a launcher: RANK / WORLD_SIZE / MASTER_* in the environment -> dist_env brings up the process
group (gloo here, nccl = RCCL on GPU boxes), the drivers shard their work, rank 0 writes the artefacts, the group is torn down.
Device kernels are replaced by the oracle-backed stand-ins of the other host tests."""
import os
import pickle

import numpy as np
import torch
import torch.multiprocessing as mp

from law_of_vision_representation_in_mllms_amd import dist_env
from test_host_ascore import _make_tree as make_ascore_tree, _no_scales, _oracle_batch
from test_host_cscore import cpu_pck_counts, cpu_transfer, make_tree as make_spair_tree
from oracle import ascore as OA


def test_no_launcher_means_no_process_group(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    assert dist_env.init_from_env() is False
    dist_env.finalize(False)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")                                # a single-process launch needs no group either
    assert dist_env.init_from_env() is False and not torch.distributed.is_initialized()


def _ascore_worker(rank, world, tmp, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from law_of_vision_representation_in_mllms_amd.A_score import compute as AC
    AC._score_batch, AC._row_scales = _oracle_batch, _no_scales
    res = AC.main(["--base-folder", tmp, "--subfolders", "encA", "encB", "--n-images", "6", "--device", "cpu"])   # stand-in score hooks: device-agnostic
    q.put((rank, res, torch.distributed.is_initialized()))


def _cscore_worker(rank, world, tmp, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.chdir(tmp)
    from law_of_vision_representation_in_mllms_amd import cscore_ops
    from law_of_vision_representation_in_mllms_amd.C_score import pck_train as PT
    cscore_ops.transfer, cscore_ops.pck_counts = cpu_transfer, cpu_pck_counts
    cfg = os.path.join(os.path.dirname(PT.__file__), "configs", "eval_zero_shot_spair.yaml")
    args = PT.parse_args(["--config", cfg, "--DATA_DIR", os.path.join(tmp, "data", "SPair-71k"), "--NOTE", "dist"])
    args.DATA_DIR, args.NOTE = os.path.join(tmp, "data", "SPair-71k"), "dist"          # the yaml's keys override the command line
    pcks = PT.main(args)
    q.put((rank, pcks, torch.distributed.is_initialized()))


def _run(worker, tmp, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, tmp, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(got)


def test_ascore_main_under_a_launcher(tmp_path):
    data = make_ascore_tree(str(tmp_path))
    got = _run(_ascore_worker, str(tmp_path), 36500 + os.getpid() % 2000)
    for rank, res, still_up in got:
        assert not still_up                                              # main() tore its own group down
        for enc in ("encA", "encB"):
            assert abs(res[enc] - OA.a_score(data[enc], data["clip336"], data["clip224"])[0]) < 1e-12


def test_pck_train_main_under_a_launcher(tmp_path):
    root, z = make_spair_tree(str(tmp_path))
    got = _run(_cscore_worker, str(tmp_path), 37500 + os.getpid() % 2000)
    for rank, pcks, still_up in got:
        assert not still_up
        np.testing.assert_allclose(pcks, z["eval.pck"], atol=1e-7)
    out = [os.path.join(r, f) for r, _, fs in os.walk(tmp_path / "results_spair") for f in fs if f == "result.pkl"]
    assert len(out) == 1                                                 # written once, by rank 0
    with open(out[0], "rb") as f:
        assert len(pickle.load(f)) == z["eval.pred"].shape[0]
