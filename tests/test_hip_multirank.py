"""N > 1 through the product's own entry points on a 1-GPU box: two processes share cuda:0, rendezvous over gloo.

  * bench.py --gpus 2 started BARE (no torchrun around it) must start its own two ranks and print a line that says so;
    a world size that does not match --gpus must fail, not report n_gpus: 1.
  * TTS.infer_batched in two processes (SPMD, same arguments on both ranks; the speaker / prompt material exists on rank 0
    only) must return on rank 0, sample for sample, the AudioClips ONE process returns (greedy, noise_scale 0): the GPT rows
    are placement-invariant and the vocoder batches are formed over the request-order lengths (tts.py), so the rank count
    changes who computes a batch, not what is in it.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gsv_tts_lite_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _bench(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                          timeout=900, env=e)


def test_bench_launches_its_own_ranks():
    r = _bench("--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--workload", "cb", "--requests", "16", "--slots", "4",
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist"]["world_size"] == 2 and out["dist"]["backend"] == "gloo"
    assert out["dist"]["speaker_broadcasts"] == 1 and out["dist"]["ranks_share_one_gpu"] is True
    assert out["config"]["requests_per_step"] == 32 and out["value"] > 0
    import re
    assert int(re.search(r"(\d+) samples arrived on rank 0", out["gather"]).group(1)) > 32 * 50 * 1280


def test_bench_refuses_a_world_size_mismatch():
    r = _bench("--gpus", "8", "--no-cpu-baseline", "--no-extras", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


# ------------------------------------------------------------------------------------------------ TTS.infer_batched
TEXTS = ["First sentence is here. Second one follows! And a third, longer one, to fill the queue.",
         "Another text", "Third, with a comma. Then more words here.", "Short.", "The fifth text has two parts. Here is the other.",
         "Six", "Seven is a little longer than six, is it not?"]


def _toy_frontend(text):
    ids = [1 + (ord(c) * 7) % 690 for c in text if not c.isspace()]
    return ids, {"word": list(text), "ph": [1] * len(text)}, None, text


def _make_tts(dev, with_refs):
    from gsv_tts import TTS
    tts = TTS(gpt_cache=[(1, 320), (3, 320)], sovits_cache=[50, 55], device=str(dev), dtype="float32")
    tts.load_gpt_model("synthetic://gpt?seed=1234&n_layer=3&eos_gain=2.5")
    tts.load_sovits_model("synthetic://sovits?version=v2Pro&seed=1234")
    tts.set_text_frontend(_toy_frontend)
    if with_refs:     # the reference-audio material exists where the reference-audio models ran: rank 0
        tts.cache_spk_audio("spk.wav", ge=torch.from_numpy(synth.synth_ge(0, 1024)))
        tts.cache_spk_audio("spk2.wav", ge=torch.from_numpy(synth.synth_ge(1, 1024)))
        x, y, _, _ = synth.synth_request(0, 12, 0, 30)
        tts.cache_prompt_audio("prompt.wav", "prompt text.", prompt=torch.from_numpy(y)[None], phones1=x.tolist())
    return tts


def _spk():
    return ["spk.wav", "spk2.wav", "spk.wav", {"spk.wav": 1.0, "spk2.wav": 3.0}, "spk.wav", "spk2.wav", "spk.wav"]


def _worker(rank, world, port, ret, dst):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    tts = _make_tts(dev, with_refs=(rank == 0))
    tts.gather_dst = dst
    clips = tts.infer_batched(_spk(), "prompt.wav", "prompt text.", TEXTS, top_k=1, noise_scale=0.0, cut_minlen=8, sovits_batch_size=3)
    ret[rank] = None if clips is None else [c.audio_data for c in clips]
    ret["bc%d" % rank] = tts._speaker_book.broadcasts
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dst", [0, None])
def test_infer_batched_two_ranks_equal_one_process_sample_for_sample(dst):
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    single = _make_tts(dev, True).infer_batched(_spk(), "prompt.wav", "prompt text.", TEXTS, top_k=1, noise_scale=0.0, cut_minlen=8,
                                                sovits_batch_size=3)
    assert len(single) == len(TEXTS) and all(len(c.audio_data) > 3200 for c in single)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret, dst), nprocs=world, join=True)
    assert ret["bc0"] == ret["bc1"] == 3 + 2      # prompt (tokens, phonemes, BERT rows) + two speakers, once each
    got = [ret[0]] if dst == 0 else [ret[0], ret[1]]
    if dst == 0:
        assert ret[1] is None
    for clips in got:
        assert len(clips) == len(single)
        for a, b in zip(clips, single):
            assert a.shape == b.audio_data.shape and np.array_equal(a, b.audio_data)
