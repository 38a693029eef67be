"""world_size-2 gloo tests (CPU) of the multi-GPU engine's host logic: on-demand dealing from the shared cursor,
the once-per-speaker broadcast, and the ordered gather -- driven by a fake decoder that has the product decoder's
slot-loop interface (infer_batched(..., source=, slots=)) and a per-request cost, so that dealing is really dynamic."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gsv_tts_lite_amd import engine, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class FakeDecoder:
    """slot loop with the reference's shape (t2s_model.py:633-728): B slots, a finished slot is refilled from the
    source; request i needs n_steps[i] steps and returns tokens that identify it; rank `slow` sleeps per step"""

    def __init__(self, n_steps, delay):
        self.n_steps, self.delay = n_steps, delay

    def infer_batched(self, xs, ys, berts, source=None, slots=4, **kw):
        live = {}
        for s in range(slots):
            c = source.next()
            if c is None:
                break
            live[s] = [c, self.n_steps[c]]
        pred, idx = [], []
        while live:
            time.sleep(self.delay)
            for s in list(live):
                live[s][1] -= 1
                if live[s][1] <= 0:
                    c = live[s][0]
                    pred.append(torch.full((self.n_steps[c],), c, dtype=torch.int64))
                    idx.append(c)
                    n = source.next()
                    if n is None:
                        del live[s]
                    else:
                        live[s] = [n, self.n_steps[n]]
        return pred, torch.tensor(idx, dtype=torch.int64)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 37
    steps = [int(v) for v in synth.hashed_ints("steps", n, 2, 9, 5)]
    xs = [torch.zeros(int(v), dtype=torch.int64) for v in synth.hashed_ints("lx", n, 3, 40, 5)]
    eng = engine.ContinuousBatchingEngine(FakeDecoder(steps, 0.004 if rank == 0 else 0.001), slots=4, chunk=2)
    out = eng.run(xs, xs, xs)
    taken = list(eng.last_taken)
    # a second run on the same group must use a fresh cursor
    out2 = eng.run(xs, xs, xs)
    # speakers: one broadcast per NEW key, dictionary hits afterwards
    book = engine.SpeakerBook("cpu")
    ge = torch.from_numpy(synth.synth_ge(3, 1024, 7)) if rank == 0 else None
    prompt = torch.arange(70) if rank == 0 else None
    a = book.sync("spk-a", [ge, prompt] if rank == 0 else None)
    n1 = book.broadcasts
    b = book.sync("spk-a", None)
    n2 = book.broadcasts
    c = book.sync("spk-b", [torch.ones(2, 3)] if rank == 0 else None)
    ok = (torch.equal(a[0], torch.from_numpy(synth.synth_ge(3, 1024, 7))) and torch.equal(a[1], torch.arange(70))
          and b is a and n1 == 2 and n2 == 2 and book.broadcasts == 3 and torch.equal(c[0], torch.ones(2, 3)))
    # variable-length tensors keyed by global index: all-gather form, point-to-point form (to rank 0 and to the last rank),
    # a rank that holds nothing, zero-length entries, a float payload
    mine = {i: torch.full((steps[i] if i % 5 else 0,), i, dtype=torch.int64) for i in taken}
    ex_all = eng.exchange(mine, n, dst=None)
    ex_0 = eng.exchange(mine, n, dst=0)
    ex_l = eng.exchange({i: v.float() * 0.5 for i, v in mine.items()}, n, dst=world - 1)
    only0 = eng.exchange({i: torch.arange(i + 1) for i in range(6)} if rank == 0 else {}, 6, dst=None)
    exp = [[i] * (steps[i] if i % 5 else 0) for i in range(n)]
    ok = ok and [t.tolist() for t in ex_all] == exp and [t.tolist() for t in only0] == [list(range(i + 1)) for i in range(6)]
    ok = ok and ((ex_0 is None) if rank != 0 else [t.tolist() for t in ex_0] == exp)
    ok = ok and ((ex_l is None) if rank != world - 1 else
                 (all(t.dtype == torch.float32 for t in ex_l) and [t.tolist() for t in ex_l] == [[0.5 * v for v in e] for e in exp]))
    try:
        eng.exchange({0: torch.zeros(1)}, 3, dst=None)      # index 0 from every rank, 1 and 2 from nobody
        ok = False
    except RuntimeError:
        pass
    g0 = eng.gather({i: ("r", i) for i in taken}, n)         # default: on rank 0 only
    ok = ok and ((g0 is None) if rank != 0 else g0 == [("r", i) for i in range(n)])
    ok = ok and eng.deal_batches(7) == [b for b in range(7) if b % world == rank]
    # retired cursors leave the store (rank 0 deletes after a gather it received)
    if rank == 0:
        ok = ok and not eng._cursor_keys
    ret[rank] = (taken, [t.tolist() for t in out], [t.tolist() for t in out2], ok, steps)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_dynamic_dealing_and_ordered_gather():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    t0, out0, out0b, ok0, steps = ret[0]
    t1, out1, out1b, ok1, _ = ret[1]
    n = len(steps)
    assert ok0 and ok1
    assert sorted(t0 + t1) == list(range(n)) and not set(t0) & set(t1), "every request dealt exactly once"
    for out in (out0, out1, out0b, out1b):          # all_gather: every rank holds every result, in input order
        assert [len(o) for o in out] == steps
        assert all(all(v == i for v in o) for i, o in enumerate(out))
    assert len(t1) > len(t0), "the 4x faster rank must have pulled more requests (dealing is on demand): %d vs %d" % (len(t1), len(t0))


def test_world4_every_request_once_and_in_order():
    """four ranks (the node has eight): the shared cursor deals every request exactly once, every rank ends with the full
    list in input order, the slow rank 0 pulls the fewest"""
    world, port = 4, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    steps = ret[0][4]
    n = len(steps)
    taken = [ret[r][0] for r in range(world)]
    assert all(ret[r][3] for r in range(world))
    assert sorted(sum(taken, [])) == list(range(n)) and sum(len(t) for t in taken) == n
    for r in range(world):
        for out in (ret[r][1], ret[r][2]):
            assert [len(o) for o in out] == steps and all(all(v == i for v in o) for i, o in enumerate(out))
    assert len(taken[0]) <= min(len(t) for t in taken[1:])


def test_single_process_engine_is_the_plain_slot_loop():
    steps = [3, 1, 4, 1, 5, 9, 2, 6]
    xs = [torch.zeros(k + 2, dtype=torch.int64) for k in steps]
    eng = engine.ContinuousBatchingEngine(FakeDecoder(steps, 0.0), slots=3)
    out = eng.run(xs, xs, xs)
    assert [len(o) for o in out] == steps
    assert sorted(eng.last_taken) == list(range(8))
    # longest-first order (cost = len(x)); ties by index
    assert eng.last_taken[0] == 5 and engine.lpt_order([3, 9, 9, 1]) == [1, 2, 0, 3]


def test_store_less_fallback_is_the_static_partition():
    """no store (world > 1 simulated): the source falls back to scheduler.shard_indices"""
    from gsv_tts_lite_amd import scheduler
    costs = [int(v) for v in synth.hashed_ints("c", 21, 5, 300, 9)]
    parts = []
    for r in range(4):
        eng = engine.ContinuousBatchingEngine.__new__(engine.ContinuousBatchingEngine)
        eng.world, eng.rank, eng.store, eng.chunk = 4, r, None, 2
        src = eng._source(costs)
        got = []
        while True:
            i = src.next()
            if i is None:
                break
            got.append(i)
        assert sorted(got) == scheduler.shard_indices(costs, 4, r)
        parts += got
    assert sorted(parts) == list(range(21))
