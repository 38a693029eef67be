"""GPU parity of enc_p on device (csrc/encp.h + tapgemm, bf16) against the torch restatement of
TextEncoder.infer (sovits_encoder.py), which the decode() golden fixtures pin to the reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gsv_tts_lite_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _vq(ver, seed, dev, dtype=torch.bfloat16):
    from gsv_tts_lite_amd.sovits import SynthesizerTrn
    hps = synth.sovits_hps(ver)
    vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    vq.load_state_dict(synth.sovits_weights(hps, seed=seed))
    vq.initialize_runtime(dtype, dev, [64])
    return vq


@pytest.mark.parametrize("ver", ["v2Pro", "v2"])
def test_enc_p_fp32_parity_mode_on_device(dev, ver):
    """fp32 handle: enc_p runs on device too (fp32 tapgemm + the plain fp32 attention / LayerNorm kernels of csrc/encp.h)
    and must agree with the torch restatement -- which decode.npz pins to the reference -- to fp32 summation-order
    accuracy: measured max 4e-6 on m_p / logs_p (values of ~0.8) and 4e-7 on the softmax probabilities; gates 2e-5 / 2e-6.
    Same cases as the bf16 test."""
    vq = _vq(ver, 7, dev, torch.float32)
    assert vq._voc.has_enc_p
    rng = np.random.default_rng(3)
    gin = 1024 if ver == "v2Pro" else 512
    for n_codes, P, mode in [(25, 30, "c"), (70, 41, "pf"), (3, 5, "c"), (150, 100, "slice")]:
        T = 2 * n_codes
        codes = torch.from_numpy(rng.integers(0, 1024, (1, 1, n_codes))).to(dev)
        text = torch.from_numpy(rng.integers(1, 700, (1, P))).to(dev)
        ge = torch.from_numpy(synth.synth_ge(1, gin, 7)).to(dev)
        sl = None
        if mode != "c":
            ge = torch.cat([ge.expand(-1, -1, T // 2), torch.from_numpy(synth.synth_ge(2, gin, 7)).to(dev).expand(-1, -1, T - T // 2)], 2)
        if mode == "slice":
            cut_t, cut_p = T // 2, P // 2
            sl = torch.tensor([[0, cut_p]] * cut_t + [[cut_p, P]] * (T - cut_t), device=dev)
        ge_in = vq.enc_p.ge_to512(ge) if vq.is_v2pro else ge
        with torch.inference_mode():
            q = vq._codebook_decode(vq._weights, codes)
            q = F.interpolate(q, size=q.shape[-1] * 2, mode="nearest")
            m_ref, logs_ref, _ = vq.enc_p.infer(q, text, ge_in, 1, slice_indices=sl)
            a_ref = vq.enc_p.mrte.cross_attention.attn[0].clone()
            m, logs, attn = vq._voc.enc_p(codes[0, 0], text[0], ge_in, sl)
        for got, ref, name in ((m, m_ref, "m_p"), (logs, logs_ref, "logs_p"), (attn, a_ref, "attn")):
            err = (got - ref).abs().max().item()
            print("enc_p fp32 %s %s %s: max |err| %.2e" % (ver, mode, name, err))
            assert got.shape == ref.shape and torch.isfinite(got).all()
            assert err < (2e-6 if name == "attn" else 2e-5), (ver, mode, name, err)
        if sl is not None:
            assert attn[:, :cut_t, cut_p:P - 1].max().item() < 1e-30     # masked at -1e4: exp underflows to 0


@pytest.mark.parametrize("ver", ["v2Pro", "v2"])
def test_enc_p_bf16_vs_torch_restatement(dev, ver):
    """vs the fp32 torch restatement -- measured: m_p / logs_p max 0.037, mean 0.007 on |x| ~ 0.8; attn max 1.3e-3 -- and
    vs the SAME restatement with bf16 roundings at the places the device path stores bf16 (round_fn), a tighter pin.
    Lengths that are not multiples of the 32-key / 128-query tiles; broadcast and per-frame ge; the
    time-concatenated batch form with slice_indices (mrte_model.py:27-33)."""
    vq = _vq(ver, 7, dev)
    assert vq._voc.has_enc_p
    from gsv_tts_lite_amd.sovits_encoder import TextEncoder
    enc16 = TextEncoder(vq.hps_model, vq._weights, dev, round_fn=lambda t: t.to(torch.bfloat16).to(torch.float32))
    rng = np.random.default_rng(3)
    gin = 1024 if ver == "v2Pro" else 512
    for n_codes, P, mode in [(25, 30, "c"), (70, 41, "pf"), (3, 5, "c"), (150, 100, "slice")]:
        T = 2 * n_codes
        codes = torch.from_numpy(rng.integers(0, 1024, (1, 1, n_codes))).to(dev)
        text = torch.from_numpy(rng.integers(1, 700, (1, P))).to(dev)
        ge = torch.from_numpy(synth.synth_ge(1, gin, 7)).to(dev)
        sl = None
        if mode != "c":
            ge = torch.cat([ge.expand(-1, -1, T // 2), torch.from_numpy(synth.synth_ge(2, gin, 7)).to(dev).expand(-1, -1, T - T // 2)], 2)
        if mode == "slice":
            cut_t, cut_p = T // 2, P // 2
            sl = torch.tensor([[0, cut_p]] * cut_t + [[cut_p, P]] * (T - cut_t), device=dev)
        ge_in = vq.enc_p.ge_to512(ge) if vq.is_v2pro else ge
        with torch.inference_mode():
            q = vq._codebook_decode(vq._weights, codes)
            q = F.interpolate(q, size=q.shape[-1] * 2, mode="nearest")
            m_ref, logs_ref, _ = vq.enc_p.infer(q, text, ge_in, 1, slice_indices=sl)
            a_ref = vq.enc_p.mrte.cross_attention.attn[0].clone()
            m16, logs16, _ = enc16.infer(q, text, ge_in, 1, slice_indices=sl)
            m, logs, attn = vq._voc.enc_p(codes[0, 0], text[0], ge_in, sl)
        assert m.shape == m_ref.shape and attn.shape == a_ref.shape
        for got, ref, name in ((m, m_ref, "m_p"), (logs, logs_ref, "logs_p")):
            err = (got - ref).abs()
            assert torch.isfinite(got).all()
            assert err.max().item() < 0.1 and err.mean().item() < 0.02, \
                (ver, n_codes, P, mode, name, err.max().item(), err.mean().item())
        for got, ref, name in ((m, m16, "m_p"), (logs, logs16, "logs_p")):   # the rounding-matched mirror
            err = (got - ref).abs()
            print("enc_p %s %s vs bf16-rounded mirror: max %.3e mean %.3e" % (mode, name, err.max().item(), err.mean().item()))
            assert err.max().item() < 0.04 and err.mean().item() < 0.008, (ver, mode, name, err.max().item(), err.mean().item())
        ea = (attn - a_ref).abs()
        assert ea.max().item() < 6e-3 and abs(attn.sum(-1) - 1).max().item() < 1e-3, \
            (ver, n_codes, P, mode, ea.max().item())
        if sl is not None:   # frames of the first utterance never attend to the second one's phonemes (except the last column)
            assert attn[:, :cut_t, cut_p:P - 1].max().item() == 0.0


def test_decode_bf16_uses_device_enc_p_and_stays_close_to_reference(dev, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    vq = _vq("v2Pro", int(g["seed"]), dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o, attn = vq.decode(T(g["v2Pro_codes"]), T(g["v2Pro_text"]), T(g["v2Pro_ge"]), noise_scale=0.0)
    vq.native_enc_p = False
    o2, attn2 = vq.decode(T(g["v2Pro_codes"]), T(g["v2Pro_text"]), T(g["v2Pro_ge"]), noise_scale=0.0)
    ref = g["v2Pro_o"]
    for out in (o, o2):
        err = np.abs(out[0, 0].cpu().numpy() - ref)
        assert err.max() < 0.15 and err.mean() < 0.015, (err.max(), err.mean())
    assert np.abs(attn.cpu().numpy() - g["v2Pro_attn"]).max() < 6e-3
