"""GPU tests of the drop-in surface: SynthesizerTrn.decode end to end against the reference's
golden output, and TTS.infer / TTS.infer_batched through the facade with synthetic checkpoints."""
import os

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


@pytest.mark.parametrize("ver", ["v2Pro", "v2"])
def test_decode_end_to_end_matches_reference(golden_dir, dev, ver):
    from gsv_tts_lite_amd.sovits import SynthesizerTrn
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    hps = synth.sovits_hps(ver)
    vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    vq.load_state_dict(synth.sovits_weights(hps, seed=int(g["seed"])))
    vq.initialize_runtime(torch.float32, dev, [50, 55])
    assert vq.samples_per_frame == 640
    o, attn = vq.decode(_T(g[ver + "_codes"], dev), _T(g[ver + "_text"], dev), _T(g[ver + "_ge"], dev), noise_scale=0.0)
    assert o.shape == (1, 1, g[ver + "_o"].shape[0])
    np.testing.assert_allclose(o[0, 0].cpu().numpy(), g[ver + "_o"], atol=1e-3)
    assert np.abs(o[0, 0].cpu().numpy() - g[ver + "_o"]).max() < 2e-4
    np.testing.assert_allclose(attn.cpu().numpy(), g[ver + "_attn"], atol=1e-5)
    ob, _ = vq.decode(_T(g[ver + "_codes"], dev), _T(g[ver + "_text"], dev), _T(g[ver + "_ge_cat"], dev), noise_scale=0.0,
                      cuda_graph=False, slice_indices=_T(g[ver + "_pairs"], dev))
    np.testing.assert_allclose(ob[0, 0].cpu().numpy(), g[ver + "_ob"], atol=1e-3)


def _toy_frontend(text):
    ids = [1 + (ord(c) * 7) % 690 for c in text if not c.isspace()]
    return ids, {"word": list(text), "ph": [1] * len(text)}, None, text


def _make_tts(dev, dtype):
    from gsv_tts import TTS, AudioClip  # the drop-in alias
    tts = TTS(gpt_cache=[(1, 128), (1, 160), (4, 160)], sovits_cache=[50, 55], device=str(dev), dtype=dtype)
    tts.load_gpt_model("synthetic://gpt?seed=1234&n_layer=6&eos_gain=1.0")
    tts.load_sovits_model("synthetic://sovits?version=v2Pro&seed=1234")
    tts.set_text_frontend(_toy_frontend)
    tts.cache_spk_audio("spk.wav", ge=torch.from_numpy(synth.synth_ge(0, 1024)))
    x, y, _, _ = synth.synth_request(0, 12, 0, 30)
    tts.cache_prompt_audio("prompt.wav", "prompt text.", prompt=torch.from_numpy(y)[None], phones1=x.tolist())
    return tts, AudioClip


def test_tts_infer_and_infer_batched(dev, tmp_path):
    tts, AudioClip = _make_tts(dev, "float32")
    assert tts.get_gpt_list() and tts.get_sovits_list()
    clip = tts.infer("spk.wav", "prompt.wav", "prompt text.", "Hello there, this is a test", top_k=1, noise_scale=0.0)
    assert isinstance(clip, AudioClip) and clip.samplerate == 32000
    assert clip.audio_data.dtype == np.float32 and clip.audio_data.ndim == 1 and len(clip.audio_data) > 6400
    assert abs(clip.audio_len_s - len(clip.audio_data) / 32000) < 1e-9 and np.isfinite(clip.audio_data).all()
    assert np.abs(clip.audio_data).max() <= 1.0 and np.all(clip.audio_data[-6400:] == 0)
    clip.save(str(tmp_path / "a.wav"))
    assert (tmp_path / "a.wav").stat().st_size > 1000
    again = tts.infer("spk.wav", "prompt.wav", "prompt text.", "Hello there, this is a test", top_k=1, noise_scale=0.0)
    # top_k=1, noise 0: same tokens, same waveform up to torch/MIOpen's kernel choice inside enc_p
    assert clip.audio_data.shape == again.audio_data.shape
    np.testing.assert_allclose(clip.audio_data, again.audio_data, atol=1e-5)
    clips = tts.infer_batched("spk.wav", "prompt.wav", "prompt text.",
                              ["First sentence is here. Second one follows!", "Another text", "Third, with a comma."],
                              top_k=1, noise_scale=0.0, cut_minlen=8)
    assert isinstance(clips, tuple) and len(clips) == 3
    for c in clips:
        assert isinstance(c, AudioClip) and np.isfinite(c.audio_data).all() and len(c.audio_data) > 0
    with pytest.raises(NotImplementedError):
        tts.infer("other.wav", "prompt.wav", "prompt text.", "x")   # no ge cached: the ref-audio models are out of scope
    with pytest.raises(ValueError):
        tts.cache_prompt_audio("p2.wav", "", prompt=torch.zeros(1, 4, dtype=torch.int64), phones1=[1, 2])
    tts.unload_gpt_model(*tts.get_gpt_list())
    assert tts.get_gpt_list() == []


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_tts_infer_stream_chunks(dev, dtype):
    """TTS.infer_stream (TTS.py:289-504): token-mode streaming with SOLA joins.  float32 runs enc_p's streaming
    branch in the torch restatement, bfloat16 the device enc_p with the cross-fade applied to the projected
    statistics (the same function: proj is affine).  Greedy + noise 0, so the streamed audio must cover the
    same frames as the one-shot infer(): total length within a few SOLA search windows."""
    tts, AudioClip = _make_tts(dev, dtype)
    text = "Hello there, this is a streaming test"
    whole = tts.infer("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0)
    clips = list(tts.infer_stream("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0, stream_chunk=8,
                                  overlap_len=2, is_cut_text=False, debug=False))
    assert len(clips) >= 2 and all(isinstance(c, AudioClip) for c in clips)
    total = 0
    last = 0.0
    for c in clips:
        assert c.samplerate == 32000 and c.audio_data.dtype == np.float32 and np.isfinite(c.audio_data).all()
        total += len(c.audio_data)
        assert c.audio_len_s > last
        last = c.audio_len_s
    assert abs(last - total / 32000) < 1e-6
    # infer() appends 0.2 s of silence, the stream 0.4 s * 1.0 ('t' is no punctuation -> '.' appended -> x1.5)
    body_whole = len(whole.audio_data) - int(0.2 * 32000)
    body_stream = total - int(0.4 * 1.5 * 32000)
    assert abs(body_stream - body_whole) <= 320 * len(clips) + 640, (body_stream, body_whole, len(clips))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 0.2)])
def test_decode_streaming_chunks_match_the_restatement(dev, dtype, tol):
    """decode(stream_mode=True) (models.py:209-215: drop valid_start_idx frames, cross-fade overlap_len frames with the previous
    chunk's tail): the one-call library path slices / cross-fades the projected statistics, the restatement
    (oracle/sovits_encoder.py) the encoder features before `proj` as the reference does -- the same function, proj is affine.
    Three chunks of one stream with growing context, as TTS.infer_stream issues them; then a second stream must start clean."""
    from gsv_tts_lite_amd.sovits import SynthesizerTrn
    from oracle.sovits_encoder import DecodeRestatement
    hps = synth.sovits_hps("v2Pro")
    vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    vq.load_state_dict(synth.sovits_weights(hps, seed=11))
    vq.initialize_runtime(dtype, dev, [50, 55])
    ref = DecodeRestatement(vq.hps_model, vq._weights, dev, vq.flow_dec)
    rng = np.random.default_rng(5)
    codes = torch.from_numpy(rng.integers(0, 1024, (1, 1, 60))).to(dev)
    text = torch.from_numpy(rng.integers(1, 700, (1, 33))).to(dev)
    ge = _T(synth.synth_ge(0, 1024, 11), dev)
    for stream in range(2):
        vq.enc_p.y_overlap = None
        ref.enc_p.y_overlap = None
        for n_tok, start in ((20, 0), (40, 30), (60, 70)):
            o, attn = vq.decode(codes[:, :, :n_tok], text, ge, noise_scale=0.0, stream_mode=True, valid_start_idx=start, overlap_len=4)
            o2, attn2, _ = ref(codes[:, :, :n_tok], text, ge, stream_mode=True, valid_start_idx=start, overlap_len=4)
            assert o.shape == o2.shape == (1, 1, (2 * n_tok - start) * 640)
            err = (o - o2).abs()
            print("stream %d chunk %d (%s): max |diff| %.2e" % (stream, n_tok, dtype, float(err.max())))
            assert float(err.max()) < tol, (stream, n_tok, float(err.max()))
        assert tuple(vq.enc_p.y_overlap.shape) == (2 * 192, 4)


def _word_frontend(text):
    """words = latin runs or single marks; one phoneme per character; norm_text == text"""
    import re
    words = re.findall(r"[A-Za-z]+|[^\sA-Za-z]", text)
    ids = [1 + (ord(c) * 7) % 690 for w in words for c in w]
    return ids, {"word": words, "ph": [len(w) for w in words]}, None, text


def _check_subtitles(subs, text, audio_len_s):
    assert subs, "no subtitles"
    prev = 0.0
    for s in subs:
        assert set(s) >= {"text", "start_s", "end_s", "orig_idx_start", "orig_idx_end"}
        assert s["start_s"] >= prev - 1e-9
        if s["end_s"] is not None:
            assert s["end_s"] >= s["start_s"] - 1e-9
            prev = s["end_s"]
        assert 0 <= s["orig_idx_start"] < s["orig_idx_end"] <= len(text) + 1
    assert prev <= audio_len_s + 1.0


def test_tts_subtitles(dev, monkeypatch):
    """return_subtitles=True through the facade (TTS.py:250-271, 444-488, 768-852): the device alignment of the
    real enc_p attention equals the oracle's path, the audio does not change, the word timings are ordered and
    the spans index the caller's text."""
    from gsv_tts_lite_amd import subtitles as sub
    from oracle import oracle as orc
    tts, AudioClip = _make_tts(dev, "bfloat16")
    tts.set_text_frontend(_word_frontend)
    seen = []
    real = sub.viterbi_monotonic

    def spy(attn):
        out = real(attn)
        seen.append((attn.float().cpu().numpy(), out.cpu().numpy()))
        return out
    monkeypatch.setattr(sub, "viterbi_monotonic", spy)
    text = "Hello there, this is a subtitle test."
    plain = tts.infer("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0)
    clip = tts.infer("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0, return_subtitles=True)
    assert plain.subtitles == [] and len(seen) == 1
    np.testing.assert_allclose(clip.audio_data, plain.audio_data, atol=1e-5)
    a, got = seen[0]
    assert a.shape[0] == 4 and np.array_equal(got, orc.viterbi_monotonic(a))
    _check_subtitles(clip.subtitles, text, clip.audio_len_s)
    for s in clip.subtitles[:-1]:
        assert text[s["orig_idx_start"]:s["orig_idx_end"]] == s["text"]
    # streaming: cumulative chunks hand out only new words; an unfinished last word has no end yet
    seen.clear()
    clips = list(tts.infer_stream("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0, stream_chunk=8,
                                  overlap_len=2, is_cut_text=False, debug=False, return_subtitles=True))
    assert len(seen) == len(clips)
    for a, got in seen:
        assert np.array_equal(got, orc.viterbi_monotonic(a))
    assert clips[-1].subtitles and clips[-1].subtitles[-1]["end_s"] is not None
    for c in clips[:-1]:
        if c.subtitles:
            assert c.subtitles[-1]["end_s"] is None
    streamed = [s for c in clips for s in c.subtitles]
    _check_subtitles(streamed, text, clips[-1].audio_len_s)


def test_tts_batched_subtitles(dev, monkeypatch):
    """infer_batched(return_subtitles=True), TTS.py:768-852: one alignment per time-concatenated vocoder batch, the
    word timings cut the batch apart, spans are shifted into each original text.  The GPT is replaced by fixed
    token lists and the attention by a per-segment ridge so that every word is reached (random weights give no
    usable attention); decode() itself (flow + Generator on the concatenated batch) is the real one."""
    tts, AudioClip = _make_tts(dev, "bfloat16")
    tts.set_text_frontend(_word_frontend)
    t2s = next(iter(tts.gpt_models.values())).t2s_model
    vq = next(iter(tts.sovits_models.values())).vq_model
    rng = np.random.default_rng(5)

    def fake_gpt(ids, prompts, berts, **kw):
        n = len(ids)
        order = list(range(n))[::-1]                      # completion order != request order
        return [torch.from_numpy(rng.integers(0, 1024, 30 + 4 * i)).to(dev) for i in order], torch.tensor(order)
    monkeypatch.setattr(t2s, "infer_batched", fake_gpt)
    real_decode = vq.decode

    def ridge_decode(codes, text, ge, **kw):
        audio, attn = real_decode(codes, text, ge, **kw)
        pairs = kw["slice_indices"].cpu().numpy()
        syn = np.zeros(tuple(attn.shape), np.float32)
        t0 = 0
        while t0 < pairs.shape[0]:
            t1 = t0
            while t1 < pairs.shape[0] and (pairs[t1] == pairs[t0]).all():
                t1 += 1
            p0, p1 = pairs[t0]
            syn[:, t0:t1, p0:p1] = synth.synth_attn(int(p0), 4, t1 - t0, int(p1 - p0), noise=0.05)
            t0 = t1
        return audio, torch.from_numpy(syn).to(attn.device)
    monkeypatch.setattr(vq, "decode", ridge_decode)
    texts = ["First one is here. Second follows!", "Another text", "Third, with a comma."]
    clips = tts.infer_batched("spk.wav", "prompt.wav", "prompt text.", texts, top_k=1, noise_scale=0.0, cut_minlen=8,
                              sovits_batch_size=3, return_subtitles=True)
    assert len(clips) == 3
    for c, t in zip(clips, texts):
        full = t if t[-1] in ".!?," else t + "."
        _check_subtitles(c.subtitles, full, c.audio_len_s)
        words = [s["text"] for s in c.subtitles]
        assert "".join(words).replace(".", "").startswith(full.replace(" ", "").replace(".", "")[:6])
        assert abs(c.subtitles[-1]["end_s"] - c.audio_len_s) < 0.75
        for s in c.subtitles:   # spans of later segments are shifted by len(segment) as in TTS.py:852, which does
            if len(s["text"]) > 1:   # not count the blank cut_text() dropped between segments
                assert s["text"] in full[max(0, s["orig_idx_start"] - 2):s["orig_idx_end"] + 2]


def test_tts_vc_async_and_cache_management(dev):
    """infer_vc (TTS.py:871-964), the asyncio wrappers (TTS.py:966-1262) and the cache bookkeeping (TTS.py:1436-1480)"""
    import asyncio
    tts, AudioClip = _make_tts(dev, "bfloat16")
    tts.set_text_frontend(_word_frontend)
    vc = tts.infer_vc("spk.wav", "prompt.wav", "prompt text.", noise_scale=0.0)
    assert isinstance(vc, AudioClip) and vc.orig_text == "prompt text." and np.isfinite(vc.audio_data).all()
    assert len(vc.audio_data) == 30 * 2 * 640 + 6400            # the cached prompt's 30 tokens -> 60 frames, + 0.2 s
    _check_subtitles(vc.subtitles, "prompt text.", vc.audio_len_s)
    text = "Hello there, async test."
    want = tts.infer("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0)

    async def go():
        one = await tts.infer_async("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0)
        many = await tts.infer_batched_async("spk.wav", "prompt.wav", "prompt text.", [text, "Second one."], top_k=1, noise_scale=0.0)
        chunks = [c async for c in tts.infer_stream_async("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0,
                                                          stream_chunk=8, overlap_len=2, debug=False)]
        return one, many, chunks
    one, many, chunks = asyncio.run(go())
    np.testing.assert_allclose(one.audio_data, want.audio_data, atol=1e-5)
    assert len(many) == 2 and len(chunks) >= 2 and all(isinstance(c, AudioClip) for c in chunks)
    assert tts.get_spk_audio_list() == ["spk.wav"] and tts.get_prompt_audio_list() == ["prompt.wav"]
    tts.del_spk_audio("spk.wav", "missing.wav")
    tts.del_prompt_audio("prompt.wav")
    assert tts.get_spk_audio_list() == [] and tts.get_prompt_audio_list() == []


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vocoder_graph_bucket_equals_eager(dev, dtype):
    """a pass whose length equals a sovits_cache bucket replays a hipGraph of the whole flow + Generator
    (gsv_voc_flow_dec_graph, the reference's per-bucket graphs, SoVITS/models.py:322-369): same kernels, same buffers
    layout -> bit-identical to the eager pass, also on the second replay with new contents"""
    from gsv_tts_lite_amd.sovits import _VocoderNative
    hps = synth.sovits_hps("v2Pro")
    w = synth.sovits_weights(hps, seed=3, hot_path_only=True)
    v = _VocoderNative(hps["model"], {k: torch.from_numpy(a) for k, a in w.items()}, dtype, dev)
    ge = _T(synth.synth_ge(1, 1024, 3), dev)
    for rep in range(3):
        z = _T(synth.hashed_uniform("gb.z%d" % rep, (1, 192, 50), 3) * np.float32(1.3), dev)
        mask = torch.ones(1, 1, 50, device=dev)
        eager = v.flow_dec(z, mask, ge)
        graph = v.flow_dec_bucket(z, mask, ge)
        assert torch.equal(eager, graph), rep
    assert len(v._buckets) == 1


def test_decode_speed_on_device_matches_torch_encoder(dev):
    """speed != 1 (TextEncoder.infer resamples the features linearly before proj, SoVITS/models.py:217-219): the device
    path resamples the projected statistics (gsv_voc_resample_linear) and must agree with the torch restatement of
    the reference order; also the resampling kernel itself against F.interpolate"""
    import torch.nn.functional as F
    from gsv_tts_lite_amd.sovits import SynthesizerTrn
    hps = synth.sovits_hps("v2Pro")
    vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    vq.load_state_dict(synth.sovits_weights(hps, seed=11))
    vq.initialize_runtime(torch.bfloat16, dev, [50, 55])
    x = torch.randn(1, 7, 83, device=dev)
    for T_out in (64, 83, 120, 167):
        got = vq._voc.resample_linear(x, T_out)
        want = F.interpolate(x, size=T_out, mode="linear")
        assert torch.allclose(got, want, atol=1e-6), T_out
    codes = torch.randint(0, 1024, (1, 1, 40), device=dev)
    text = torch.randint(1, 700, (1, 30), device=dev)
    ge = _T(synth.synth_ge(0, 1024, 11), dev)
    from oracle.sovits_encoder import DecodeRestatement
    ref = DecodeRestatement(vq.hps_model, vq._weights, dev, vq.flow_dec)
    for speed in (1.25, 0.8):
        o1, _ = vq.decode(codes, text, ge, noise_scale=0.0, speed=speed)
        o2, _, _ = ref(codes, text, ge, speed=speed)
        assert o1.shape == o2.shape == (1, 1, (int(80 / speed) + 1) * 640)
        err = (o1 - o2).abs()
        assert err.max() < 0.15 and err.mean() < 0.02, (speed, float(err.max()), float(err.mean()))   # bf16 enc_p vs fp32 torch enc_p
