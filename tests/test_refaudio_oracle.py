"""CPU: the reference-audio oracle (oracle.spectrogram / RefAudioOracle) against the reference's outputs in
tests/golden/refaudio.npz (SynthesizerTrn.get_ge / extract_latent of the imported reference; |torch.stft| for the
torchaudio Spectrogram of TTS._get_spec).  Tolerances: spectrogram 2e-4 of its peak, ge 2e-5 abs (fp32 restatement
of fp32 torch), codes bit-exact."""
import os

import numpy as np
import pytest

from gsv_tts_lite_amd import synth
from oracle import oracle as orc

CASES = [("v2Pro", 0, 32000 * 3 + 123, 151), ("v2", 1, 40000, 64), ("v2ProPlus", 2, 2048, 3)]


@pytest.mark.parametrize("ver,i,n_samples,n_ssl", CASES)
def test_refaudio_oracle_matches_reference(golden_dir, ver, i, n_samples, n_ssl):
    g = np.load(os.path.join(golden_dir, "refaudio.npz"))
    hps = synth.sovits_hps(ver)
    o = orc.RefAudioOracle(synth.ref_audio_weights(hps, int(g["seed"])))
    spec = orc.spectrogram(synth.synth_audio(i, n_samples))
    assert spec.shape == (1025, 1 + n_samples // 640)
    want = g[ver + "_spec_sub"]
    np.testing.assert_allclose(spec[::8, ::4], want, atol=2e-4 * want.max(), rtol=0)
    assert abs(spec.astype(np.float64).sum() - float(g[ver + "_spec_sum"])) < 1e-5 * float(g[ver + "_spec_sum"])
    sv = synth.synth_sv_emb(i) if ver != "v2" else None
    ge = o.get_ge(spec, sv)
    np.testing.assert_allclose(ge, g[ver + "_ge"][0, :, 0], atol=2e-5, rtol=0)
    codes, margin = o.extract_latent(synth.synth_ssl(i, n_ssl)[0])
    assert np.array_equal(codes, g[ver + "_codes"][0, 0]) and margin.min() > 0
