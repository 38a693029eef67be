"""GPU parity of enc_p on device (csrc/encp.h + tapgemm) and of the one-call decode (gsv_voc_decode) against the torch
restatement of TextEncoder.infer / SynthesizerTrn.decode (oracle/sovits_encoder.py: test infrastructure), which the
decode() golden fixtures pin to the reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gsv_tts_lite_amd import synth, _native as N

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
    from oracle.sovits_encoder import TextEncoder, DecodeRestatement, codebook_decode
    vq.ref_enc = TextEncoder(vq.hps_model, vq._weights, dev)
    vq.ref_decode = DecodeRestatement(vq.hps_model, vq._weights, dev, vq.flow_dec)
    vq.ref_codebook = codebook_decode
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
        ge_in = vq.ref_enc.ge_to512(ge) if vq.is_v2pro else ge
        with torch.inference_mode():
            q = vq.ref_codebook(vq._weights, codes)
            q = F.interpolate(q, size=q.shape[-1] * 2, mode="nearest")
            m_ref, logs_ref, _ = vq.ref_enc.infer(q, text, ge_in, 1, slice_indices=sl)
            a_ref = vq.ref_enc.mrte.cross_attention.attn[0].clone()
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
    from oracle.sovits_encoder import TextEncoder
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
        ge_in = vq.ref_enc.ge_to512(ge) if vq.is_v2pro else ge
        with torch.inference_mode():
            q = vq.ref_codebook(vq._weights, codes)
            q = F.interpolate(q, size=q.shape[-1] * 2, mode="nearest")
            m_ref, logs_ref, _ = vq.ref_enc.infer(q, text, ge_in, 1, slice_indices=sl)
            a_ref = vq.ref_enc.mrte.cross_attention.attn[0].clone()
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
    o2, attn2, _ = vq.ref_decode(T(g["v2Pro_codes"]), T(g["v2Pro_text"]), T(g["v2Pro_ge"]))      # fp32 torch enc_p -> bf16 flow_dec
    ref = g["v2Pro_o"]
    for out in (o, o2):
        err = np.abs(out[0, 0].cpu().numpy() - ref)
        assert err.max() < 0.15 and err.mean() < 0.015, (err.max(), err.mean())
    assert np.abs(attn.cpu().numpy() - g["v2Pro_attn"]).max() < 6e-3


def test_decode_noise_is_the_counter_based_stream_and_per_token_ge_is_an_index_map(dev):
    """gsv_voc_decode end to end in the fp32 parity mode against the restatement: (i) noise_scale != 0 with the library's noise
    (oracle.device_normal restates it: z_p = m_p + n * exp(logs_p) * noise_scale, models.py:404), replayable from the generator
    seed; (ii) per-TOKEN ge of a time-concatenated batch (x2 nearest upsampling, models.py:389) with slice_indices and speed != 1
    (nearest resize to the resampled length, :402)."""
    from oracle import oracle as orc
    vq = _vq("v2Pro", 7, dev, torch.float32)
    rng = np.random.default_rng(9)
    n, P = 37, 29
    codes = torch.from_numpy(rng.integers(0, 1024, (1, 1, n))).to(dev)
    text = torch.from_numpy(rng.integers(1, 700, (1, P))).to(dev)
    ge = torch.from_numpy(synth.synth_ge(1, 1024, 7)).to(dev)
    g = torch.Generator(device="cpu"); g.manual_seed(77)
    seed = int(torch.empty((), dtype=torch.int64).random_(generator=g).item()) & (2 ** 64 - 1)   # what decode() will draw first
    g.manual_seed(77)
    o1, _ = vq.decode(codes, text, ge, noise_scale=0.5, generator=g)
    noise = torch.from_numpy(orc.device_normal(seed, 192 * 2 * n).astype(np.float32)).reshape(1, 192, 2 * n).to(dev)
    assert abs(float(noise.mean())) < 0.03 and abs(float(noise.std()) - 1.0) < 0.03
    o_ref, _, _ = vq.ref_decode(codes, text, ge, noise=noise, noise_scale=0.5)
    e = (o1 - o_ref).abs()
    print("decode with noise vs restatement + device_normal: max %.2e" % float(e.max()))
    assert float(e.max()) < 1e-4
    o1c, _ = vq.decode(codes, text, ge, noise_scale=0.5, generator=g)
    assert not torch.equal(o1, o1c), "the next call of a run draws fresh noise: the generator's state advanced"
    g.manual_seed(77)                                  # re-seeding replays the run, as torch.randn(generator=...) would
    o1b, _ = vq.decode(codes, text, ge, noise_scale=0.5, generator=g)
    o1d, _ = vq.decode(codes, text, ge, noise_scale=0.5, generator=g)
    assert torch.equal(o1, o1b) and torch.equal(o1c, o1d), "same generator seed, same call index: the same draw"
    g2 = torch.Generator(device="cpu"); g2.manual_seed(77)
    o1e, _ = vq.decode(codes, text, ge, noise_scale=0.5, generator=g2)
    assert torch.equal(o1, o1e), "a second generator with the same seed starts the same stream (no shared counter)"
    # per-token ge: two utterances concatenated, each with its own speaker
    cut = 15
    ge2 = torch.from_numpy(synth.synth_ge(2, 1024, 7)).to(dev)
    ge_cat = torch.cat([ge.expand(-1, -1, cut), ge2.expand(-1, -1, n - cut)], 2)
    sl = torch.tensor([[0, 12]] * (2 * cut) + [[12, P]] * (2 * (n - cut)), device=dev)
    for speed in (1.0, 1.3):
        o, attn = vq.decode(codes, text, ge_cat, noise_scale=0.0, speed=speed, cuda_graph=False, slice_indices=sl)
        o_r, attn_r, _ = vq.ref_decode(codes, text, ge_cat, speed=speed, slice_indices=sl)
        assert o.shape == o_r.shape
        e = (o - o_r).abs()
        print("per-token ge, speed %.1f: max %.2e" % (speed, float(e.max())))
        assert float(e.max()) < 1e-4 and float((attn - attn_r).abs().max()) < 1e-5
    with pytest.raises(ValueError):
        vq.decode(codes.expand(2, -1, -1), text, ge)       # batched codes: the reference never passes them, the library refuses


def test_decode_frame_count_is_the_callers_at_every_speed(dev):
    """int(T / speed) + 1 (models.py:217) is evaluated ONCE, in Python doubles as the reference does, and handed to
    the library; a float32 re-evaluation inside disagreed at e.g. (speed 1.1, T 110) and (0.6, 9).  Sweep chunk lengths at those
    speeds: the output has exactly the reference's length, every sample is written (no NaN canary left), nothing is written
    past the end, and the graph-bucket path agrees with the eager one."""
    vq = _vq("v2Pro", 7, dev, torch.float32)
    rng = np.random.default_rng(3)
    P = 11
    text = torch.from_numpy(rng.integers(1, 700, (1, P))).to(dev)
    ge = torch.from_numpy(synth.synth_ge(1, 1024, 7)).to(dev)
    hop = vq.samples_per_frame
    cases = [(55, 1.1, 0), (5, 0.6, 1), (50, 1.1, 0), (7, 0.6, 0), (12, 0.7, 3), (55, 2.0, 0), (33, 0.9, 0)]   # (n, speed, valid_start)
    import torch as _t
    real_empty = _t.empty
    for n, speed, start in cases:
        Tp = 2 * n - start
        T_ref = int(Tp / speed) + 1
        codes = torch.from_numpy(rng.integers(0, 1024, (1, 1, n))).to(dev)
        kw = dict(noise_scale=0.0, speed=speed)
        if start:
            vq.enc_p.y_overlap = None
            kw.update(stream_mode=True, valid_start_idx=start, overlap_len=2)
        o, _ = vq.decode(codes, text, ge, cuda_graph=False, **kw)
        assert o.shape[-1] == T_ref * hop, (n, speed, start, o.shape, T_ref)
        assert bool(torch.isfinite(o).all())
        o_r, _, _ = vq.ref_decode(codes, text, ge, speed=speed) if not start else (None, None, None)
        if o_r is not None:
            assert o.shape == o_r.shape and float((o - o_r).abs().max()) < 1e-4
        # the same call into a NaN-filled, guard-padded buffer through the ABI: all of [0, T_ref * hop) written, nothing after
        vn = vq._voc
        c1 = codes.reshape(-1).contiguous(); t1 = text.reshape(-1).contiguous()
        g1 = ge.to(torch.float32).reshape(vq.gin_channels, -1).contiguous()
        need = N.lib().gsv_voc_decode_workspace(vn._h, n, P, 1, T_ref, start)
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        buf = torch.full((T_ref * hop + 4 * hop,), float("nan"), device=dev)
        state = torch.zeros(2 * vq.inter_channels, 2, device=dev)
        N.check(N.lib().gsv_voc_decode(vn._h, c1.data_ptr(), n, t1.data_ptr(), P, g1.data_ptr(), 1, 0, 0.0, 0, T_ref, start,
                                       2 if start else 0, state.data_ptr() if start else 0, 0, 0, buf.data_ptr(), 0, ws.data_ptr(), ws.numel(),
                                       N.current_stream_ptr(dev)))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(buf[:T_ref * hop]).all()) and bool(torch.isnan(buf[T_ref * hop:]).all()), (n, speed, start)
        assert torch.equal(buf[:T_ref * hop], o.reshape(-1))
    # a bucketed (hipGraph) chunk at speed != 1
    vq.cuda_graph_buckets = [int(110 / 1.1) + 1]
    codes = torch.from_numpy(rng.integers(0, 1024, (1, 1, 55))).to(dev)
    a, _ = vq.decode(codes, text, ge, noise_scale=0.0, speed=1.1, cuda_graph=True)
    b, _ = vq.decode(codes, text, ge, noise_scale=0.0, speed=1.1, cuda_graph=False)
    assert a.shape == b.shape and torch.equal(a, b)
    assert N.lib().gsv_voc_decode_workspace(vq._voc._h, 10, P, 1, 0, 0) == 0      # out_frames < 1 is refused


def test_vocoder_graph_cache_evicts_instead_of_failing(dev):
    """the captured-pass cache is bounded (GSV_VOC_MAX_GRAPHS = 64) and evicts its least recently replayed entry; a
    long-lived server that keeps meeting new (workspace, T) pairs must keep decoding."""
    from gsv_tts_lite_amd.sovits import _VocoderNative
    hps = synth.sovits_hps("v2Pro")
    sw = synth.sovits_weights(hps, seed=5, hot_path_only=True)
    voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.bfloat16, dev)
    ge = torch.from_numpy(synth.synth_ge(0, 1024, 5)).to(dev)
    first = None
    for k in range(70):                       # 70 distinct lengths -> 70 captures through a 64-entry cache
        T = 4 + k
        z = torch.from_numpy(synth.hashed_uniform("gc.z", (1, 192, T), 5)).to(dev)
        o = voc.flow_dec_bucket(z, torch.ones(1, 1, T, device=dev), ge)
        assert bool(torch.isfinite(o).all())
        if k == 0:
            first = (z, o.clone())
    z, o0 = first                              # evicted by now: re-captured, same result
    assert torch.equal(voc.flow_dec_bucket(z, torch.ones(1, 1, 4, device=dev), ge), o0)
