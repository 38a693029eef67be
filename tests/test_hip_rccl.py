"""RCCL itself, on the one GPU a test box has: `torch.distributed` backend "nccl" (= RCCL on ROCm) initialised with ONE rank,
and every collective the multi-GPU path issues run through it -- so that `bench.py --gpus 8` on the driver's node cannot die in
RCCL initialisation or in a collective nobody ever executed.

engine.FORCE_COLLECTIVES (GSV_FORCE_COLLECTIVES=1) makes a one-rank process group take the backend's code path instead of the
single-process short cuts: the store cursor (`store.add`), `SpeakerBook.sync` (broadcast_object_list + broadcast),
`ContinuousBatchingEngine.exchange` (all-reduce of the table, the dtype agreement, the padded all-gather; with dst = 0 the
point-to-point branch), `scheduler.max_over_ranks`.  RCCL's send / recv between two different ranks needs two devices; here the
p2p API is exercised as a self send/recv inside one batch, which is what RCCL supports on one rank.

  * reference for what is being distributed: the slot queue of Text2SemanticDecoder.infer_batched
    (gsv_tts/GPT_SoVITS/GPT/t2s_model.py:696-722) and TTS.infer_batched's vocoder batches (gsv_tts/TTS.py:616-633,705-764).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _env(**extra):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
              "GSV_FORCE_COLLECTIVES": "1", "PYTHONPATH": os.pathsep.join([ROOT, os.path.join(ROOT, "gsv-tts-lite_amd"), e.get("PYTHONPATH", "")])})
    e.update(extra)
    return e


_PROBE = r'''
import json, os, sys
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from gsv_tts_lite_amd import engine, scheduler
assert engine.FORCE_COLLECTIVES and engine._dist_on()
res = {"backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")}

# -- speakers: one broadcast per tensor of a new key, then a dictionary hit
book = engine.SpeakerBook(dev)
ge = torch.arange(1024, dtype=torch.float32).reshape(1, 1024, 1)
tok = torch.arange(30, dtype=torch.int64)
got = book.sync("spk", [ge, tok])
again = book.sync("spk", None)
res["broadcasts"] = book.broadcasts
res["speaker_ok"] = bool(got[0].is_cuda and torch.equal(got[0].cpu(), ge) and torch.equal(got[1].cpu(), tok) and again is got)

# -- exchange: variable-length device tensors keyed by request index
class _Dec:          # exchange() never calls the decoder
    device = dev
eng = engine.ContinuousBatchingEngine(_Dec(), slots=4, chunk=2)
assert eng.world == 1 and eng.store is not None and not eng._staged()
local = {i: torch.arange(3 + 5 * i, device=dev, dtype=torch.int64) + 100 * i for i in range(6)}
every = eng.exchange(local, 6, dst=None)          # all-reduce (table) + all-reduce MAX (dtype) + all-gather (padded)
res["allgather_ok"] = bool(all(t.is_cuda and torch.equal(t, local[i]) for i, t in enumerate(every)))
audio = {i: torch.randn(640 * (i + 1), device=dev) for i in range(6)}
on0 = eng.exchange(audio, 6, dst=0)               # the point-to-point branch (own buffer: no wire at one rank)
res["gather_dst0_ok"] = bool(all(torch.equal(t, audio[i]) for i, t in enumerate(on0)))
res["empty_rank_ok"] = True
try:
    eng.exchange({0: audio[0]}, 2, dst=None)      # a missing request is reported, not returned as garbage
    res["missing_detected"] = False
except RuntimeError:
    res["missing_detected"] = True

# -- the shared request cursor on the process group's store
src = eng._source([5, 9, 1, 7, 3])
order = []
while True:
    i = src.next()
    if i is None:
        break
    order.append(i)
res["cursor_order"] = order
res["cursor_via_store"] = src.store is not None
eng._retire_cursors(None)

# -- timing reduction and the barrier bench.py brackets its timed region with
res["max_over_ranks"] = scheduler.max_over_ranks(1.25, device=dev)
dist.barrier()
t = torch.ones(1 << 20, device=dev)
dist.all_reduce(t)
res["allreduce_ok"] = bool(float(t.sum()) == float(1 << 20))

# -- RCCL point-to-point API: self send/recv in one batch (two ranks need two devices)
try:
    a, b = torch.arange(4096, device=dev, dtype=torch.float32), torch.zeros(4096, device=dev)
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):
        w.wait()
    torch.cuda.synchronize()
    res["p2p_self"] = bool(torch.equal(a, b))
except Exception as exc:
    res["p2p_self"] = "unsupported: %r" % (exc,)
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
'''


def test_rccl_one_rank_runs_every_collective_of_the_engine():
    r = subprocess.run([sys.executable, "-c", _PROBE], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(res)
    assert res["backend"] == "nccl (RCCL)"
    assert res["broadcasts"] == 2 and res["speaker_ok"]
    assert res["allgather_ok"] and res["gather_dst0_ok"] and res["missing_detected"]
    assert res["cursor_via_store"] and res["cursor_order"] == [1, 3, 0, 4, 2]       # longest first, through store.add
    assert res["max_over_ranks"] == 1.25 and res["allreduce_ok"]
    assert res["p2p_self"] is True or str(res["p2p_self"]).startswith("unsupported"), res["p2p_self"]


def test_bench_cb_runs_over_rccl_at_world_size_one():
    """`bench.py --workload cb` with the process group forced on: the slot loop pulls from the store cursor, token ids go through
    the all-gather, samples through the dst = 0 exchange, the timed region through barrier + max_over_ranks -- all on RCCL."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cb", "--requests", "24", "--slots", "8", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["dist"]["backend"] == "nccl (RCCL)" and out["dist"]["world_size"] == 1 and out["dist"]["speaker_broadcasts"] == 1
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["requests_per_step"] == 24
    import re
    assert int(re.search(r"(\d+) samples arrived on rank 0", out["gather"]).group(1)) > 24 * 50 * 1280
