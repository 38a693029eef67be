"""BASELINE configs[2] and configs[4] at FULL size on one MI355X, as property runs (the oracle finishes such a queue in hours,
so the small-size tests in test_hip_t2s_lowp.py compare with it and this file checks what is size-independent):

  configs[2]  V2ProPlus continuous batching, 32 slots, 256 mixed-length requests, 24 layers, bf16
  configs[4]  the same queue through 64 slots with e4m3 QKV / FFN operands in the batched step

  * every request is served exactly once (the global index set is 0..255), whatever order the slots finish in;
  * a request's token count equals its budget (EOS weight zero: the budget is the only terminator, t2s_model.py:680-694);
  * tokens are valid semantic ids (< 1024: EOS never appears in an output);
  * decoding is placement-invariant: the same queue through 8 more slots returns the same tokens for every request; the
    matched-prefix statistics against the bs = 1 kernels and against the fp32 parity mode are printed;
  * the vocoder stage as TTS.infer_batched runs it (TTS.py:705-764): length-balanced batches of 10, every utterance's samples
    finite, of length 2 * tokens * 640, non-silent.
"""
import numpy as np
import pytest
import torch

from gsv_tts_lite_amd import synth

pytestmark = pytest.mark.gpu
N_REQ = 256


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def queue(dev):
    lens = synth.mixed_lengths(N_REQ)
    budgets = synth.mixed_new_tokens(N_REQ)
    reqs = [synth.synth_request(i, 40, t, n) for i, (t, n) in enumerate(lens)]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return [T(r[0]) for r in reqs], [T(r[1]) for r in reqs], [T(r[2]) for r in reqs], budgets


@pytest.fixture(scope="module")
def fp32_tokens(dev, queue):
    """the fp32 parity mode on a sample of the queue, each request alone (its tokens are the oracle's: test_hip_bench_size.py)"""
    from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
    xs, ys, bs, budgets = queue
    cfg = synth.gpt_config()
    m = Text2SemanticDecoder(cfg)
    m.load_state_dict(synth.gpt_weights(cfg, seed=1234, eos_gain=0.0))
    m.initialize_runtime(torch.float32, dev, [(1, 512), (1, 1024)])
    out = {}
    for i in range(0, N_REQ, 16):
        out[i] = m.infer(xs[i][None], ys[i][None], bs[i][None], top_k=1, max_new_tokens=budgets[i])[0, 0].cpu().numpy()
    del m
    torch.cuda.empty_cache()
    return out


def _prefix(a, b):
    n = min(len(a), len(b))
    d = np.nonzero(np.asarray(a[:n]) != np.asarray(b[:n]))[0]
    return n if d.size == 0 else int(d[0])


@pytest.mark.parametrize("which,dtype,slots", [("configs[2]", torch.bfloat16, 32), ("configs[4]", torch.float8_e4m3fn, 64)])
def test_full_size_queue(dev, queue, fp32_tokens, which, dtype, slots):
    from gsv_tts_lite_amd import engine
    from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
    from gsv_tts_lite_amd.sovits import _VocoderNative
    from gsv_tts_lite_amd.batchmath import balance_order
    xs, ys, bs, budgets = queue
    cfg = synth.gpt_config()
    m = Text2SemanticDecoder(cfg)
    m.load_state_dict(synth.gpt_weights(cfg, seed=1234, eos_gain=0.0))
    m.initialize_runtime(dtype, dev, [(1, 512), (1, 1024), (slots, 512), (slots, 1024), (slots + 8, 512), (slots + 8, 1024)])
    assert slots >= m.batched_min, "this batch size must run the batched MFMA chain"
    eng = engine.ContinuousBatchingEngine(m, slots=slots, chunk=2)
    pred, idx = eng.run_gpt(xs, ys, bs, top_k=1, max_new_tokens=budgets)
    assert sorted(idx.tolist()) == list(range(N_REQ)), "every request exactly once"
    by_req = {int(i): p.cpu().numpy() for i, p in zip(idx.tolist(), pred)}
    for i in range(N_REQ):
        assert len(by_req[i]) == budgets[i], (i, len(by_req[i]), budgets[i])
        assert by_req[i].min() >= 0 and by_req[i].max() < 1024
    st = m.last_stats
    tok = sum(budgets)
    print("%s: %d tokens in %d slot-loop steps of %d slots: %.1f %% of slot-steps idle, %d refills"
          % (which, tok, st["steps"], slots, 100.0 * (1.0 - tok / (st["steps"] * slots)), st["refills"]))
    assert st["steps"] * slots >= tok and st["refills"] == N_REQ - slots

    # placement invariance: the same queue through a DIFFERENT number of slots (other rows of other tiles, other refill times,
    # another completion order) returns, request by request, the same tokens -- rows are independent through every kernel
    # of the chain and a request's prompt pass does not depend on what it was packed with
    other = slots + 8
    eng2 = engine.ContinuousBatchingEngine(m, slots=other, chunk=2)
    pred2, idx2 = eng2.run_gpt(xs, ys, bs, top_k=1, max_new_tokens=budgets)
    assert sorted(idx2.tolist()) == list(range(N_REQ))
    bad = [int(i) for i, p in zip(idx2.tolist(), pred2) if not np.array_equal(p.cpu().numpy(), by_req[int(i)])]
    assert not bad, "requests %s decode differently in %d slots than in %d" % (bad[:8], other, slots)

    # distance from the bs = 1 kernels and from the fp32 parity mode on a sample of the queue: REPORTED.  These random-weight models
    # have nearly flat logits (top-1 / top-2 gaps of ~0.3 on logits of O(6)), so reduced precision flips a decision every few
    # steps and the runs part early; what the arithmetic must hold is pinned against the rounding-matched oracle at sizes it can
    # run (test_hip_t2s_lowp.py), not here.  Only chance-level agreement (1 / 1024 per token) would mean a wrong K/V row or slot.
    alone, vs32, first_ok = [], [], 0
    for i in sorted(fp32_tokens):
        t1 = m.infer(xs[i][None], ys[i][None], bs[i][None], top_k=1, max_new_tokens=budgets[i])[0, 0].cpu().numpy()
        alone.append(_prefix(t1, by_req[i]))
        vs32.append(_prefix(fp32_tokens[i], by_req[i]))
        first_ok += int(fp32_tokens[i][0] == by_req[i][0])
    print("%s: matched prefix in tokens, slot loop vs the same request alone (bs = 1 kernels): %s; slot loop vs fp32 parity mode: %s; "
          "first token equal to the parity mode's in %d of %d sampled requests" % (which, alone, vs32, first_ok, len(vs32)))
    assert first_ok >= len(vs32) // 4, (first_ok, vs32)
    del m

    # the vocoder stage of TTS.infer_batched: V2ProPlus, length-balanced batches of 10, per-frame ge
    hps = synth.sovits_hps("v2ProPlus")
    sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
    voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.bfloat16, dev)
    ge = torch.from_numpy(synth.synth_ge(0, hps["model"]["gin_channels"], 1234)).to(dev)
    lengths = torch.tensor([budgets[i] for i in range(N_REQ)])
    order = balance_order(lengths)
    seen, frames = set(), 0
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    for s in range(0, N_REQ, 10):
        batch = order[s:s + 10].tolist()
        T = int(sum(2 * int(lengths[i]) for i in batch))
        z = torch.randn(1, 192, T, device=dev, generator=gen)
        o = voc.flow_dec(z, torch.ones(1, 1, T, device=dev), ge.expand(-1, -1, T).contiguous())[0, 0]
        assert o.numel() == T * voc.samples_per_frame
        pos = 0
        for i in batch:
            n_ = 2 * int(lengths[i]) * voc.samples_per_frame
            a = o[pos:pos + n_]
            pos += n_
            assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 1e-4 and float(a.abs().max()) <= 1.0, i
            seen.add(i)
        frames += T
    assert seen == set(range(N_REQ)) and frames == 2 * tok
    print("%s: vocoder stage: %d frames (%.0f s of audio) in 26 batches, every utterance finite and non-silent" % (which, frames, frames / 50.0))
