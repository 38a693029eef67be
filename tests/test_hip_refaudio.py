"""GPU parity of the reference-audio path (csrc/refaudio.h behind gsv_ref_*): spectrogram, get_ge, extract_latent
against the reference's outputs (tests/golden/refaudio.npz) and against the oracle on further lengths.
Tolerances (fp32 MFMA path vs fp32 torch CPU): spectrogram 3e-4 of its peak, ge 1e-4 abs on |ge| ~ 0.7,
codes bit-exact wherever the oracle's best-vs-second distance gap exceeds 1e-2 (the distances are ~1e3)."""
import os

import numpy as np
import pytest
import torch

from gsv_tts_lite_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

CASES = [("v2Pro", 0, 32000 * 3 + 123, 151), ("v2", 1, 40000, 64), ("v2ProPlus", 2, 2048, 3)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _vq(ver, dev, seed=1234):
    from gsv_tts_lite_amd.sovits import SynthesizerTrn
    hps = synth.sovits_hps(ver)
    vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    sd = dict(synth.sovits_weights(hps, seed=seed, hot_path_only=True))
    sd.update(synth.ref_audio_weights(hps, seed=seed))
    vq.load_state_dict(sd)
    vq.initialize_runtime(torch.bfloat16, dev, [50])
    return vq, hps


@pytest.mark.parametrize("ver,i,n_samples,n_ssl", CASES)
def test_refaudio_golden(dev, golden_dir, ver, i, n_samples, n_ssl):
    g = np.load(os.path.join(golden_dir, "refaudio.npz"))
    vq, hps = _vq(ver, dev, int(g["seed"]))
    spec = vq.spectrogram(torch.from_numpy(synth.synth_audio(i, n_samples)))
    assert spec.shape == (1, 1025, 1 + n_samples // 640)
    want = g[ver + "_spec_sub"]
    np.testing.assert_allclose(spec[0].cpu().numpy()[::8, ::4], want, atol=3e-4 * want.max(), rtol=0)
    sv = torch.from_numpy(synth.synth_sv_emb(i)) if ver != "v2" else None
    ge = vq.get_ge(spec, sv)
    assert ge.shape == g[ver + "_ge"].shape
    np.testing.assert_allclose(ge.cpu().numpy(), g[ver + "_ge"], atol=1e-4, rtol=0)
    ssl = synth.synth_ssl(i, n_ssl)
    codes = vq.extract_latent(torch.from_numpy(ssl))
    assert codes.dtype == torch.int64 and codes.shape == g[ver + "_codes"].shape
    _, margin = orc.RefAudioOracle(synth.ref_audio_weights(hps, int(g["seed"]))).extract_latent(ssl[0])
    ok = margin > 1e-2
    assert ok.mean() > 0.9
    assert np.array_equal(codes[0, 0].cpu().numpy()[ok], g[ver + "_codes"][0, 0][ok])


def test_refaudio_vs_oracle_lengths(dev):
    """frame counts around the 64-row GEMM tiles, odd ssl lengths (the stride-2 conv drops the last frame), ge
    without sv_emb on a v2Pro model (models.py:374 skips the tail when sv_emb is None)"""
    vq, hps = _vq("v2Pro", dev)
    o = orc.RefAudioOracle(synth.ref_audio_weights(hps, 1234))
    for k, n in enumerate([1025, 640 * 63, 640 * 64 + 1, 640 * 129 + 639]):
        a = synth.synth_audio(10 + k, n)
        spec = vq.spectrogram(torch.from_numpy(a)[None])
        so = orc.spectrogram(a)
        np.testing.assert_allclose(spec[0].cpu().numpy(), so, atol=3e-4 * so.max(), rtol=0)
        sv = synth.synth_sv_emb(10 + k) if k % 2 == 0 else None
        ge = vq.get_ge(spec, None if sv is None else torch.from_numpy(sv))
        np.testing.assert_allclose(ge[0, :, 0].cpu().numpy(), o.get_ge(so, sv), atol=1e-4, rtol=0)
    for k, n in enumerate([2, 3, 127, 128, 129, 500]):
        ssl = synth.synth_ssl(20 + k, n)
        ref = vq._ref_audio()
        codes, margin = ref.extract_latent(torch.from_numpy(ssl), return_margin=True)
        want, wm = o.extract_latent(ssl[0])
        ok = wm > 1e-2
        assert codes.shape == (1, 1, n // 2)
        assert np.array_equal(codes[0, 0].cpu().numpy()[ok], want[ok])
        np.testing.assert_allclose(margin.cpu().numpy(), wm, atol=5e-3)


def test_tts_caches_from_audio_and_ssl(dev):
    """the facade computes ge from a waveform (+ sv_emb) and the prompt tokens from ssl features on the device"""
    from gsv_tts import TTS
    tts = TTS(gpt_cache=[(1, 128), (1, 160)], sovits_cache=[50], device=str(dev), dtype="bfloat16")
    tts.load_gpt_model("synthetic://gpt?seed=1234&n_layer=4&eos_gain=1.0")
    tts.load_sovits_model("synthetic://sovits?version=v2Pro&seed=1234")
    tts.set_text_frontend(lambda t: ([1 + (ord(c) * 7) % 690 for c in t if not c.isspace()], {"word": list(t), "ph": [1] * len(t)}, None, t))
    a = synth.synth_audio(3, 32000 * 2)
    tts.cache_spk_audio("spk.wav", audio=torch.from_numpy(a) * 4.0, sv_emb=torch.from_numpy(synth.synth_sv_emb(3)))   # peak > 1: rescaled
    ge = next(iter(tts.spk_audio_cache["spk.wav"]["ge"].values()))
    hps = synth.sovits_hps("v2Pro")
    o = orc.RefAudioOracle(synth.ref_audio_weights(hps, 1234))
    a4 = a * np.float32(4.0)
    a4 = a4 / np.float32(min(2.0, float(np.abs(a4).max())))
    np.testing.assert_allclose(ge[0, :, 0].cpu().numpy(), o.get_ge(orc.spectrogram(a4), synth.synth_sv_emb(3)), atol=2e-4, rtol=0)
    ssl = synth.synth_ssl(3, 60)
    x, _, _, _ = synth.synth_request(0, 12, 0, 30)
    tts.cache_prompt_audio("prompt.wav", "prompt text.", ssl_content=torch.from_numpy(ssl), phones1=x.tolist())
    prompt = tts.prompt_audio_cache["prompt.wav"]["prompt"]
    assert prompt.shape == (1, 30) and prompt.dtype == torch.int64
    clip = tts.infer("spk.wav", "prompt.wav", "prompt text.", "Hello there", top_k=1, noise_scale=0.0)
    assert np.isfinite(clip.audio_data).all() and len(clip.audio_data) > 6400
    with pytest.raises(NotImplementedError):
        tts.cache_spk_audio("other.wav")
