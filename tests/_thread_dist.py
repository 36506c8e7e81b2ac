"""TEST INFRASTRUCTURE: a torch.distributed stand-in that runs `world` ranks as THREADS of one process on one device.

Why: the sweep's C leg moves every launch's maps to the rank that owns their category with an owner-addressed `all_to_all_single`
(sweep.c_score_of).  On the 1-GPU boxes this repo is developed on that exchange had only ever moved CPU tensors of stand-in towers
(gloo, world 2): RCCL refuses two ranks on one GPU and gloo has no device all-to-all.  Here the collectives are rendezvous on a
threading.Barrier and the payload is copied between REAL device buffers with the REAL split tables - everything but the transport
is the production code path (index tables, send / receive splits, bank rows, asynchronous completion order).

The ranks are COOPERATIVE: a rank thread holds one run lock while it computes and gives it up only inside a collective, so between two
collectives exactly one rank runs - like one process per rank, each rank's launches reach the (shared) stream as one uninterrupted
sequence.  (Free-running threads on one stream are not what the production launcher does, and they break an assumption every torch
program makes: a temporary whose pointer was handed to a launch may be freed before the launch is enqueued, because the same thread's
next allocation can only be written by a LATER kernel of the same stream - another thread's allocation can slip in between.)

Only the calls sweep.py / C_score.pck_train make are implemented: get_rank, get_world_size, barrier, all_to_all_single (sync or
async_op), all_gather_object, all_reduce (SUM), all_gather.
"""
import threading

import torch


class _Handle:
    def __init__(self, fn):
        self._fn, self._done = fn, False

    def wait(self):
        if not self._done:
            self._fn()
            self._done = True


class ThreadDist:
    def __init__(self, world: int):
        self.world = world
        self._tls = threading.local()
        self._bar = threading.Barrier(world)
        self._run = threading.Lock()                  # held by the one rank that is computing
        self._slots = [None] * world
        self.bytes_on_fabric = 0                      # rows that changed rank, in bytes (all ranks)
        self._lock = threading.Lock()

    # ---- what torch.distributed exposes
    def is_available(self):
        return True

    def is_initialized(self):
        return True

    def get_rank(self):
        return self._tls.rank

    def get_world_size(self):
        return self.world

    def _wait(self):
        """rendezvous: give the run lock up while waiting, take it back before computing again"""
        self._run.release()
        try:
            self._bar.wait()
        finally:
            self._run.acquire()

    def barrier(self):
        self._wait()

    def _exchange(self, obj):
        """every rank deposits obj; returns the list of all ranks' objects (valid until the next collective)"""
        r = self.get_rank()
        self._wait()
        self._slots[r] = obj
        self._wait()
        return list(self._slots)

    def all_gather_object(self, out_list, obj):
        got = self._exchange(obj)
        for i in range(self.world):
            out_list[i] = got[i]

    def all_reduce(self, t, op=None):
        got = self._exchange(t.clone())
        t.copy_(sum(g.to(t.device) for g in got))

    def all_gather(self, parts, t, async_op=False):
        got = self._exchange(t)
        def fin():
            for i in range(self.world):
                parts[i].copy_(got[i])
        if async_op:
            return _Handle(fin)
        fin()
        return None

    def all_to_all_single(self, out, inp, out_splits, in_splits, async_op=False):
        """out = concat over source ranks r of the rows r sends to me; inp rows are grouped by destination rank (in_splits)."""
        me = self.get_rank()
        got = self._exchange((inp, list(in_splits)))                   # rendezvous: every rank's send buffer + its split table

        def fin():
            pos = 0
            for r in range(self.world):
                src, splits = got[r]
                off = sum(splits[:me])
                n = splits[me]
                assert n == out_splits[r], (me, r, n, out_splits[r])   # the receiver's table must agree with the sender's
                if n:
                    out[pos:pos + n].copy_(src[off:off + n])            # device-to-device
                    if r != me:
                        with self._lock:
                            self.bytes_on_fabric += src[off:off + n].numel() * src.element_size()
                pos += n
            assert pos == out.shape[0]
        if async_op:
            return _Handle(fin)
        fin()
        return None

    # ---- driver
    def run(self, fn):
        """fn(rank) on `world` threads; returns the list of results, re-raises the first failure"""
        res, err = [None] * self.world, []

        def body(r):
            self._tls.rank = r
            self._run.acquire()
            try:
                res[r] = fn(r)
            except BaseException as e:                                  # noqa: BLE001 - surfaced below
                err.append(e)
                self._bar.abort()
            finally:
                self._run.release()
        th = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            raise err[0]
        return res
