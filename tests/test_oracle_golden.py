"""CPU tests: the C/numpy oracle against the golden vectors produced by the imported
reference (oracle/gen_golden.py).  This is what pins the oracle:
the reference ships no tests of its own, so every expected value below is the reference's
actual output on seeded inputs, generated in the build container.

Tolerances: the oracle sums in a different order than torch's CPU kernels, so float
tensors agree to fp32 round-off (<= 2e-5 abs on O(1) activations); greedy token ids
must be identical (recorded top-1/top-2 margins are >= 3e-3, far above that noise).
"""
import os

import numpy as np
import pytest

from gsv_tts_lite_amd import synth
from oracle import oracle as orc

ATOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_sampling_matches_reference(golden_dir):
    g = _load(golden_dir, "sample.npz")
    for i in range(6):
        top_k, top_p, temp, rep = g["c%d_kw" % i]
        prev = g["c%d_prev" % i] if ("c%d_prev" % i) in g.files else None
        kw = dict(top_k=int(top_k), top_p=float(top_p), temperature=float(temp), repetition_penalty=float(rep))
        lg = g["c%d_logits" % i].copy()
        probs = orc.logits_to_probs(lg.copy(), prev, **kw)
        np.testing.assert_allclose(probs, g["c%d_probs" % i], atol=1e-6)
        idx = orc.sample(lg.copy(), prev, q=g["c%d_q" % i], **kw)
        assert np.array_equal(idx, g["c%d_idx" % i])


def test_t2s_layers_match_reference(golden_dir):
    g = _load(golden_dir, "t2s_layers.npz")
    cfg = synth.gpt_config(n_layer=3)
    o = orc.T2SOracle(cfg, synth.gpt_weights(cfg, seed=int(g["seed"])), [(1, 96), (2, 96)])
    x, y, bert = g["s_x"], g["s_y"], g["s_bert"]
    xy = np.concatenate([o.embed_text(x, bert), o.embed_audio(y)])[None]
    np.testing.assert_allclose(xy, g["s_xy"], atol=ATOL)
    m = o.single_mask(len(x), len(y))
    assert np.array_equal(m, g["s_mask"])
    L = len(x) + len(y)
    h = o.prefill(xy, m[None], 1, 0)
    np.testing.assert_allclose(h, g["s_hidden"], atol=ATOL)
    kc, vc = o.cache[1]
    np.testing.assert_allclose(kc[:, 0, :, :L], g["s_k"], atol=ATOL)
    np.testing.assert_allclose(vc[:, 0, :, :L], g["s_v"], atol=ATOL)
    hd = o.decode(g["d_x"][0], 1, [L])
    np.testing.assert_allclose(hd, g["d_hidden"][0], atol=ATOL)
    np.testing.assert_allclose(kc[:, 0, :, L], g["d_k_new"], atol=ATOL)
    # packed batch: rows [x_b | y_b | pad]; padded query rows are fully masked -> zeros in, finite out
    xs = [g["b0_x"], g["b1_x"]]; ys = [g["b0_y"], g["b1_y"]]; bs = [g["b0_bert"], g["b1_bert"]]
    Lmax = g["b_xy"].shape[1]
    bxy = np.zeros((2, Lmax, 512), np.float32)
    bm = np.zeros((2, Lmax, Lmax), np.uint8)
    for b in range(2):
        lx, ly = len(xs[b]), len(ys[b])
        bxy[b, :lx] = o.embed_text(xs[b], bs[b]); bxy[b, lx:lx + ly] = o.embed_audio(ys[b])
        bm[b, :lx + ly, :lx + ly] = o.single_mask(lx, ly)
    np.testing.assert_allclose(bxy, g["b_xy"], atol=ATOL)
    assert np.array_equal(bm, g["b_mask"])
    bh = o.prefill(bxy, bm, 2, 0)
    for b in range(2):  # only valid rows are defined; the reference's padded rows are don't-care
        n = len(xs[b]) + len(ys[b])
        np.testing.assert_allclose(bh[b, :n], g["b_hidden"][b, :n], atol=ATOL)
        assert g["b_last"][b, n - 1] == 1 and g["b_last"][b].sum() == 1
    assert np.isfinite(bh).all()


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_greedy_infer_tokens_bit_exact(golden_dir, name):
    g = _load(golden_dir, "t2s_infer.npz")
    seed, p, t, n = (int(v) for v in g[name + "_cfg"])
    cfg = synth.gpt_config()
    w = synth.gpt_weights(cfg, seed=seed, eos_gain=float(g[name + "_eos_gain"]))
    o = orc.T2SOracle(cfg, w, [tuple(int(v) for v in c) for c in g[name + "_cache"]])
    x, y = g[name + "_x"], g[name + "_y"]
    tok = o.infer(x, y, np.zeros((len(x), 1024), np.float32), top_k=1)
    assert np.array_equal(tok, g[name + "_tokens"])
    np.testing.assert_allclose(np.array(o.raw_margins, np.float32), g[name + "_margins"], atol=2e-4)


@pytest.mark.parametrize("name", ["r", "s"])
def test_greedy_infer_batched_matches_reference(golden_dir, name):
    """continuous batching incl. slot refill (case r: 5 requests through 2 slots), completion
    order and semantic_orig_idx (t2s_model.py:679-681,733)."""
    g = _load(golden_dir, "t2s_batched.npz")
    seed = int(g[name + "_seed"])
    cfg = synth.gpt_config()
    w = synth.gpt_weights(cfg, seed=seed, eos_gain=float(g[name + "_eos_gain"]))
    o = orc.T2SOracle(cfg, w, [tuple(int(v) for v in c) for c in g[name + "_cache"]])
    rs = [synth.synth_request(200 + i, int(p), int(t), int(n), seed=seed) for i, (p, t, n) in enumerate(g[name + "_reqs"])]
    pred, orig = o.infer_batched([r[0] for r in rs], [r[1] for r in rs], [r[2] for r in rs], top_k=1)
    assert np.array_equal(orig, g[name + "_orig"])
    assert len(pred) == int(g[name + "_n"])
    for i, pt in enumerate(pred):
        assert np.array_equal(pt, g["%s_tok%d" % (name, i)]), i


def test_vocoder_flow_and_generator_match_reference(golden_dir):
    g = _load(golden_dir, "vocoder.npz")
    for ver, T, tag in [("v2Pro", 50, "c"), ("v2Pro", 55, "pf"), ("v2ProPlus", 50, "c"), ("v2", 23, "c"),
                        ("v2Pro", 200, "c"), ("v2ProPlus", 55, "pf")]:
        hps = synth.sovits_hps(ver)
        v = orc.VocoderOracle(hps, synth.sovits_weights(hps, seed=int(g["seed"])))
        name = "%s_T%d_%s" % (ver, T, tag)
        z, ge = g[name + "_z"][0], g[name + "_ge"][0]
        mask = np.ones(T, np.float32)
        zf = v.flow(z, mask, ge)
        np.testing.assert_allclose(zf, g[name + "_flow"][0], atol=ATOL)
        o = v.flow_dec(z, mask, ge)
        assert o.shape == (T * 640,)
        if name + "_o" in g:
            np.testing.assert_allclose(o, g[name + "_o"], atol=5e-5)
        else:   # long cases store every 5th sample and the fp64 sum
            np.testing.assert_allclose(o[::5], g[name + "_o_sub"], atol=5e-5)
            assert abs(float(o.astype(np.float64).sum()) - float(g[name + "_o_sum"])) < 0.5


def test_synth_generator_is_stable():
    """the hashed generator must give bit-identical tensors everywhere: pin a few values."""
    u = synth.hashed_uniform("pin", (4,), 1234)
    assert u.dtype == np.float32
    again = synth.hashed_uniform("pin", (4,), 1234)
    assert np.array_equal(u, again)
    assert not np.array_equal(u, synth.hashed_uniform("pin", (4,), 1235))
    ints = synth.hashed_ints("pin", 8, 0, 1024, 1234)
    assert ints.min() >= 0 and ints.max() < 1024
    np.testing.assert_array_equal(ints, synth.hashed_ints("pin", 8, 0, 1024, 1234))


@pytest.mark.parametrize("name", ["e", "f"])
def test_greedy_infer_stream_chunks_match_reference(golden_dir, name):
    """t2s_model.py:466-553: every (cumulative chunk, is_final) pair, incl. the final chunk after an EOS break
    that still carries the first sample."""
    g = _load(golden_dir, "t2s_stream.npz")
    seed, p, t, n, chunk, boost = (int(v) for v in g[name + "_cfg"])
    cfg = synth.gpt_config()
    w = synth.gpt_weights(cfg, seed=seed, eos_gain=float(g[name + "_eos_gain"]))
    o = orc.T2SOracle(cfg, w, [tuple(int(v) for v in c) for c in g[name + "_cache"]])
    x, y = g[name + "_x"], g[name + "_y"]
    got = list(o.infer_stream(x, y, np.zeros((len(x), 1024), np.float32), top_k=1, stream_chunk=chunk, boost_first_chunk=bool(boost)))
    assert len(got) == int(g[name + "_n"])
    for i, (c, fin) in enumerate(got):
        assert np.array_equal(c, g["%s_chunk%d" % (name, i)]) and int(fin) == int(g["%s_final%d" % (name, i)]), (name, i)
