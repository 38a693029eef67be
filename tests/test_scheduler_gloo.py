"""world_size-2 gloo test of the multi-GPU logic on CPU: utterance sharding, the speaker-embedding
broadcast (the only collective of the hot path) and the host-side result gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gsv_tts_lite_amd import scheduler, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = [int(v) for v in synth.hashed_ints("len", 11, 20, 400, 7)]
    mine = scheduler.shard_indices(lengths, world, rank)
    ge = torch.from_numpy(synth.synth_ge(0, 1024, 7)) if rank == 0 else torch.zeros(1, 1024, 1)
    prompt = torch.arange(50) if rank == 0 else torch.zeros(50, dtype=torch.int64)
    scheduler.broadcast_speaker([ge, prompt], src=0)
    ok = bool(torch.equal(ge, torch.from_numpy(synth.synth_ge(0, 1024, 7))) and torch.equal(prompt, torch.arange(50)))
    local = [(i, "utt%d@rank%d" % (i, rank)) for i in mine]
    merged = scheduler.gather_objects(local, dst=0)
    t = scheduler.max_over_ranks(1.0 + rank)
    ret[rank] = (mine, ok, merged, t)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_a_balanced_partition():
    lengths = [int(v) for v in synth.hashed_ints("len", 37, 20, 400, 3)]
    for world in (1, 2, 4, 8):
        parts = [scheduler.shard_indices(lengths, world, r) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lengths)
    assert scheduler.shard_indices([], 4, 1) == []


def test_world2_gloo_broadcast_and_gather():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    mine0, ok0, merged0, t0 = ret[0]
    mine1, ok1, merged1, t1 = ret[1]
    assert ok0 and ok1
    assert sorted(mine0 + mine1) == list(range(11)) and not set(mine0) & set(mine1)
    assert merged1 is None and len(merged0) == 11
    assert [m.split("@")[0] for m in merged0] == ["utt%d" % i for i in range(11)]
    assert t0 == 2.0 and t1 == 2.0
