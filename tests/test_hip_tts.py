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
    if dtype == "bfloat16":   # device enc_p streaming vs the torch streaming branch: same chunking, close audio
        vq = next(iter(tts.sovits_models.values())).vq_model
        assert vq._voc.has_enc_p
        vq.native_enc_p = False
        clips2 = list(tts.infer_stream("spk.wav", "prompt.wav", "prompt text.", text, top_k=1, noise_scale=0.0, stream_chunk=8,
                                       overlap_len=2, is_cut_text=False, debug=False))
        assert len(clips2) == len(clips)
        assert abs(sum(len(c.audio_data) for c in clips2) - total) <= 320 * len(clips)
