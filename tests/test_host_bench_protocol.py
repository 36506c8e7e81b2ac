"""bench.py's measurement protocol on 2 gloo ranks (CPU): warm-up steps are not timed, exactly K steps are, both fences hold
every rank, the reported time is the MAX over ranks and the value is the whole-job aggregate."""
import os
import time

import torch.multiprocessing as mp

import bench


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(time.perf_counter())
        time.sleep(0.02 if rank == 0 else 0.06)                     # rank 1 is three times slower
        return len(calls)
    dt, last = bench.timed_steps(step, steps=5, warmup=2, dist=dist, device=None)
    q.put((rank, dt, last, len(calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_timing_is_max_over_ranks_and_counts_exactly_k_steps():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (dt, last, n) for r, dt, last, n in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (dt0, last0, n0), (dt1, last1, n1) = got[0], got[1]
    assert n0 == n1 == 7 and last0 == last1 == 7                       # 2 warm-up + 5 timed steps on every rank
    assert dt0 == dt1                                                  # the all-reduced maximum
    assert 5 * 0.06 <= dt0 < 5 * 0.06 + 0.25                           # the slow rank's five steps, not the warm-up, not the fast rank's
    assert bench.aggregate_value(2, 256, 5, dt0) == 2 * 256 * 5 / dt0


def test_single_process_protocol_needs_no_process_group():
    n = []
    dt, last = bench.timed_steps(lambda: n.append(1) or len(n), steps=3, warmup=1)
    assert last == 4 and len(n) == 4 and dt >= 0
