"""Parity at the size bench.py measures (BASELINE configs[1]), in the fp32 parity mode, and the facade against oracle tokens.

  * GPT: 24 layers, buckets [(1,512),(1,1024)], 100 phonemes + 100 prompt tokens, 250 greedy tokens (kv 200 -> 450): token ids
    BIT-EXACT against the fp32 CPU oracle (north star) -- t2s_model.py:385-464.
  * vocoder: flow + Generator at T = 500 frames against the fp32 oracle, waveform within 1e-3 (north star) --
    SoVITS/models.py:380-383.
  * TTS.infer / TTS.infer_batched (TTS.py:150-286, 507-868): the returned audio against a composition, written out here, of
    ORACLE tokens -> SynthesizerTrn.decode (pinned by decode.npz) -> the split / trim / mute arithmetic (pinned by facade.npz).
"""
import numpy as np
import pytest
import torch

from gsv_tts_lite_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_fp32_greedy_tokens_at_the_bench_shape_are_bit_exact(dev):
    from oracle import oracle as orc
    from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
    cfg = synth.gpt_config()
    w = synth.gpt_weights(cfg, seed=1234, eos_gain=0.0)
    m = Text2SemanticDecoder(cfg)
    m.load_state_dict(w)
    m.initialize_runtime(torch.float32, dev, [(1, 512), (1, 1024)])       # bench.py GPT_CACHE
    for i in (0, 1):
        x, y, bert, _ = synth.synth_request(i, 40, 60, 100, seed=1234)
        o = orc.T2SOracle(cfg, w, [(1, 256), (1, 450)])      # the oracle stops when its cache is full: 250 tokens
        ref = o.infer(x, y, bert, top_k=1)
        tok = m.infer(_T(x, dev)[None], _T(y, dev)[None], _T(bert, dev)[None], top_k=1, max_new_tokens=250)[0, 0].cpu().numpy()
        assert len(ref) == 250 and len(tok) == 250
        neq = np.nonzero(tok != ref)[0]
        mm = np.asarray(o.margins)
        print("request %d: min top-1/top-2 logit gap over 251 decisions %.3e; first mismatch %s" % (i, mm.min(), neq[:1]))
        # BOTH requests, all 250 tokens: the north star's "bit-exact token ids from greedy AR decode" at the benchmark's own
        # shape (smallest oracle margin on these two requests: 1.07e-3, two hundred times the fp32 summation-order noise)
        assert neq.size == 0, "request %d: tokens differ first at step %d (oracle margin there %.3e)" % (i, int(neq[0]), mm[int(neq[0]) + 1])


def test_fp32_flow_dec_at_500_frames_within_1e3_of_the_oracle(dev):
    from oracle import oracle as orc
    from gsv_tts_lite_amd.sovits import _VocoderNative
    T = 500
    for ver in ("v2Pro", "v2ProPlus"):
        hps = synth.sovits_hps(ver)
        w = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
        v = _VocoderNative(hps["model"], {k: torch.from_numpy(a) for k, a in w.items()}, torch.float32, dev)
        z = synth.hashed_uniform("bench.z", (1, 192, T), 1234) * np.float32(1.2)
        ge = synth.synth_ge(0, 1024, 1234)
        ref = orc.VocoderOracle(hps, w).flow_dec(z[0], np.ones(T, np.float32), ge[0])
        out = v.flow_dec(_T(z, dev), torch.ones(1, 1, T, device=dev), _T(ge, dev))[0, 0].cpu().numpy()
        assert out.shape == ref.shape == (T * 640,)
        e = np.abs(out - ref)
        print("%s flow_dec fp32 T=500 vs oracle: max %.2e mean %.2e (rms of the waveform %.2f)" % (ver, e.max(), e.mean(), np.sqrt((ref ** 2).mean())))
        assert e.max() < 1e-3          # the north star's bound
        assert e.max() < 5e-5, e.max()  # what the kernels hold (measured 6e-6 at T <= 200)
        del v


# ------------------------------------------------------------------------------------------------ facade vs oracle tokens
SEED = 1234
PROMPT_PH, PROMPT_TOK = 12, 30


def _toy_frontend(text):
    ids = [1 + (ord(c) * 7) % 690 for c in text if not c.isspace()]
    return ids, {"word": list(text), "ph": [1] * len(text)}, None, text


def _make_tts(dev, gpt_cache, n_layer, eos_gain):
    from gsv_tts import TTS
    tts = TTS(gpt_cache=gpt_cache, sovits_cache=[50, 55], device=str(dev), dtype="float32")
    tts.load_gpt_model("synthetic://gpt?seed=%d&n_layer=%d&eos_gain=%s" % (SEED, n_layer, eos_gain))
    tts.load_sovits_model("synthetic://sovits?version=v2Pro&seed=%d" % SEED)
    tts.set_text_frontend(_toy_frontend)
    tts.cache_spk_audio("spk.wav", ge=torch.from_numpy(synth.synth_ge(0, 1024)))
    x, y, _, _ = synth.synth_request(0, PROMPT_PH, 0, PROMPT_TOK)
    tts.cache_prompt_audio("prompt.wav", "prompt text.", prompt=torch.from_numpy(y)[None], phones1=x.tolist())
    return tts, x, y


def _oracle(gpt_cache, n_layer, eos_gain):
    from oracle import oracle as orc
    cfg = synth.gpt_config(n_layer=n_layer)
    return orc.T2SOracle(cfg, synth.gpt_weights(cfg, seed=SEED, eos_gain=eos_gain), gpt_cache)


def test_tts_infer_audio_equals_oracle_tokens_through_decode(dev):
    cache = [(1, 128), (1, 192)]
    tts, x1, y = _make_tts(dev, cache, 6, 1.0)
    text = "Hello there, this is a test."
    clip = tts.infer("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0)
    ph2 = _toy_frontend(text)[0]
    x = np.asarray(x1.tolist() + ph2, np.int64)
    # infer() is deterministic also when the utterance ends by filling the largest bucket (t2s_model.py:425), as this one may
    tok = _oracle(cache, 6, 1.0).infer(x, y, np.zeros((len(x), 1024), np.float32), top_k=1, repetition_penalty=1.35)
    assert len(tok) > 8
    vq = next(iter(tts.sovits_models.values())).vq_model
    ge = tts.spk_audio_cache["spk.wav"]["ge"][next(iter(tts.sovits_models))]
    o, _ = vq.decode(_T(tok, dev)[None, None], _T(np.asarray(ph2, np.int64), dev)[None], ge, noise_scale=0.0)
    a = o[0, 0]
    a = a[tts._find_head_threshold_offsets(a):].float().cpu().numpy()        # TTS.py:265-267
    peak = np.abs(a).max()
    if peak > 1:
        a = a / peak
    a = np.concatenate([a, np.zeros(int(0.2 * 32000), np.float32)])          # TTS.py:282
    assert clip.audio_data.shape == a.shape, (clip.audio_data.shape, a.shape, len(tok))
    e = np.abs(clip.audio_data - a).max()
    print("TTS.infer vs oracle tokens -> decode: %d tokens, max |diff| %.2e" % (len(tok), e))
    assert e < 1e-4


def test_tts_infer_batched_audio_equals_oracle_tokens_through_batched_decode(dev):
    """TTS.py:616-633 (segment list), :705-764 (length-balanced, time-concatenated vocoder batches with per-frame ge and
    slice_indices), :806-816 (split + trim), :820-865 (silence, per-text concatenation) written out over the oracle's tokens."""
    from gsv_tts_lite_amd.batchmath import balance_order, split_bounds
    slots = 3
    cache = [(1, 320), (slots, 320)]
    tts, x1, y = _make_tts(dev, cache, 4, 2.0)
    texts = ["First sentence is here.", "Another text, with a comma.", "Third!", "Number four is longer than the others are.",
             "Five.", "Six is the last but one?", "Seven"]
    BS = 3
    clips = tts.infer_batched("spk.wav", "prompt.wav", "prompt text.", texts, top_k=1, noise_scale=0.0, is_cut_text=False, sovits_batch_size=BS)
    segs = [t if t[-1] in ".!?," else t + "." for t in texts]            # TTS.py:613 (no cutting: one segment per text)
    ph2 = [_toy_frontend(s)[0] for s in segs]
    xs = [np.asarray(x1.tolist() + p, np.int64) for p in ph2]
    o = _oracle(cache, 4, 2.0)
    pred, idx = o.infer_batched(xs, [y] * len(xs), [np.zeros((len(x), 1024), np.float32) for x in xs], top_k=1)
    tokens = [None] * len(xs)
    for p, i in zip(pred, np.asarray(idx).tolist()):
        tokens[i] = np.asarray(p, np.int64)
    # a request that ends by filling the cache is cut where the 5-step cadence falls, which the staged refill does not keep
    # (tests/test_hip_engine.py): this comparison needs EOS-terminated requests
    assert all(4 < len(t) < 320 - len(x) - len(y) - 8 for t, x in zip(tokens, xs)), [len(t) for t in tokens]
    vq = next(iter(tts.sovits_models.values())).vq_model
    ge = tts.spk_audio_cache["spk.wav"]["ge"][next(iter(tts.sovits_models))].squeeze(0)     # [gin, 1]
    lengths = torch.tensor([len(t) for t in tokens])
    order = balance_order(lengths).tolist()
    want = [None] * len(xs)
    for s in range(0, len(order), BS):
        oi = order[s:s + BS]
        ln = [int(lengths[i]) for i in oi]
        ge_cat = torch.cat([ge.expand(-1, l) for l in ln], dim=1)[None]
        ph_cat = _T(np.concatenate([np.asarray(ph2[i], np.int64) for i in oi]), dev)[None]
        ends = np.cumsum([len(ph2[i]) for i in oi])
        pairs = np.stack([ends - np.asarray([len(ph2[i]) for i in oi]), ends], 1)
        sl = _T(np.repeat(pairs, [2 * l for l in ln], axis=0).astype(np.int64), dev)
        audio, _ = vq.decode(_T(np.concatenate([tokens[i] for i in oi]), dev)[None, None], ph_cat, ge_cat, noise_scale=0.0,
                             cuda_graph=False, slice_indices=sl)
        audio = audio[0, 0]
        peak = audio.abs().max()
        if peak > 1.0:
            audio = audio / peak
        for i, (lo, hi) in zip(oi, split_bounds(ln, 640, 1.0)):
            a = audio[lo:hi]
            h, t = tts._find_head_threshold_offsets(a), tts._find_tail_threshold_offsets(a)
            want[i] = a[h:-t].float().cpu().numpy()
    mute = {".": 1.5, "!": 1.5, "?": 1.5, ",": 1.0}
    for k, c in enumerate(clips):
        a = np.concatenate([want[k], np.zeros(int(0.4 * mute[segs[k][-1]] * 32000), np.float32)])
        assert c.audio_data.shape == a.shape, (k, c.audio_data.shape, a.shape)
        assert np.abs(c.audio_data - a).max() < 1e-4, (k, np.abs(c.audio_data - a).max())
