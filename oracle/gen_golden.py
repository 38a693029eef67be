"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz by running the IMPORTED REFERENCE.

Runs only in the build container (needs /root/reference).  The reference source never
enters the repo: what is committed is this script plus the small input/output vectors it
writes.  Weights are NOT stored -- they are regenerated from a seed by
gsv_tts_lite_amd.synth on every box (bit-identical by construction).

    python oracle/gen_golden.py            # rewrites tests/golden/

Each fixture records torch version / thread count (the reference's numbers are tied to
this torch build's CPU kernels).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "gsv-tts-lite_amd"))

from ref_harness import import_reference, reference_functions, reference_statements  # noqa: E402
from gsv_tts_lite_amd import synth  # noqa: E402

import tqdm  # noqa: E402
import functools  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
META = dict(torch_version=torch.__version__, threads=torch.get_num_threads())


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build_ref_gpt(T2S, cfg, weights, gpt_cache):
    m = T2S(cfg)
    m.load_state_dict({k: tt(v) for k, v in weights.items()})
    m.eval()
    m.initialize_runtime(torch.float32, torch.device("cpu"), gpt_cache)
    # the reference allocates its KV roots with torch.empty: stale NaNs times a zero
    # probability poison SDPA outputs.  Zero them so the fixtures are reproducible.
    with torch.inference_mode():
        for b, bks in m.cuda_graph_buckets.items():
            bks[-1].k_cache.zero_()
            bks[-1].v_cache.zero_()
            bks[-1].kv_cache_len.zero_()
    return m


def with_margins(m):
    margins = []
    orig = m.ar_predict_layer.forward

    def hook(h):
        l = orig(h)
        t = torch.topk(l, 2, dim=-1).values
        margins.append((t[..., 0] - t[..., 1]).min().item())
        return l

    m.ar_predict_layer.forward = hook
    return margins


def gen_t2s_layers(T2S):
    """Per-layer numerics: prefill (single + packed batch) and one decode step, 3 layers."""
    cfg = synth.gpt_config(n_layer=3)
    w = synth.gpt_weights(cfg, seed=11)
    m = build_ref_gpt(T2S, cfg, w, [(1, 96), (2, 96)])
    out = {}
    with torch.inference_mode():
        x, y, bert, _ = synth.synth_request(0, 9, 14, 14, seed=11, bert="random")
        xy, mask = m.process_single_data(tt(x)[None], tt(y)[None], tt(bert)[None])
        bk = m.cuda_graph_buckets[1][-1]
        h = m.t2s_transformer.process_prompt(xy, bk.k_cache, bk.v_cache, bk.kv_cache_len, mask)
        out.update(s_x=x, s_y=y, s_bert=bert, s_xy=xy.numpy(), s_mask=mask[0, 0].numpy().astype(np.uint8),
                   s_hidden=h.numpy(), s_k=bk.k_cache[:, 0, :, :37].numpy().copy(),
                   s_v=bk.v_cache[:, 0, :, :37].numpy().copy())
        # one decode step on top of that prefill
        L = xy.shape[1]
        xin = tt(synth.hashed_uniform("dec.x", (1, 1, 512), 11))
        bk.decode_attn_mask.fill_(False)
        bk.decode_attn_mask[:, :, :, : L + 1] = True
        hd = m.t2s_transformer.decode_next_token(xin, bk.k_cache, bk.v_cache, bk.kv_cache_len,
                                                 bk.decode_attn_mask, bk.batch_indices)
        out.update(d_x=xin.numpy(), d_hidden=hd.numpy(), d_k_new=bk.k_cache[:, 0, :, L].numpy().copy())
        # packed batch of two rows with different lengths
        reqs = [synth.synth_request(1, 7, 10, 12, seed=11, bert="random"),
                synth.synth_request(2, 9, 17, 15, seed=11, bert="random")]
        from torch.nn.utils.rnn import pad_sequence
        bx = pad_sequence([tt(r[0]) for r in reqs], batch_first=True)
        by = pad_sequence([tt(r[1]) for r in reqs], batch_first=True)
        bb = pad_sequence([tt(r[2]) for r in reqs], batch_first=True)
        xl = torch.tensor([[len(r[0])] for r in reqs]); yl = torch.tensor([[len(r[1])] for r in reqs])
        bxy, last, bmask = m.process_batch_data(bx, by, bb, xl, yl)
        bk2 = m.cuda_graph_buckets[2][-1]
        bh = m.t2s_transformer.process_prompt(bxy, bk2.k_cache, bk2.v_cache, bk2.kv_cache_len, bmask)
        for i, r in enumerate(reqs):
            out["b%d_x" % i], out["b%d_y" % i], out["b%d_bert" % i] = r[0], r[1], r[2]
        out.update(b_xy=bxy.numpy(), b_mask=bmask[:, 0].numpy().astype(np.uint8), b_hidden=bh.numpy(),
                   b_last=last.numpy().astype(np.uint8))
    np.savez_compressed(os.path.join(GOLD, "t2s_layers.npz"), seed=11, **META, **out)
    print("t2s_layers ok")


def gen_t2s_infer(T2S):
    cfg = synth.gpt_config()
    out = {}
    cases = [  # name, weight seed, eos_gain, (prompt_ph, text_ph, prompt_tok), buckets
        ("a", 1234, 1.0, (12, 18, 25), [(1, 96), (1, 128)]),
        ("b", 1234, 6.0, (10, 22, 30), [(1, 200)]),
        ("c", 21, 1.0, (40, 60, 100), [(1, 256), (1, 288)]),
    ]
    for name, seed, eg, (p, t, n), cache in cases:
        w = synth.gpt_weights(cfg, seed=seed, eos_gain=eg)
        m = build_ref_gpt(T2S, cfg, w, cache)
        margins = with_margins(m)
        x, y, bert, _ = synth.synth_request(100 + len(out), p, t, n, seed=seed)
        with torch.inference_mode():
            tok = m.infer(tt(x)[None], tt(y)[None], tt(bert)[None], top_k=1)
        out.update({name + "_x": x, name + "_y": y, name + "_tokens": tok[0, 0].numpy(),
                    name + "_margins": np.array(margins, np.float32),
                    name + "_cfg": np.array([seed, p, t, n], np.int64), name + "_eos_gain": eg,
                    name + "_cache": np.array(cache, np.int64)})
        print("infer case", name, "tokens", tok.shape[-1], "min margin", min(margins),
              "distinct", len(set(tok[0, 0].tolist())))
    np.savez_compressed(os.path.join(GOLD, "t2s_infer.npz"), **META, **out)


def gen_t2s_stream(T2S):
    """greedy infer_stream() (t2s_model.py:466-553): every (cumulative chunk, is_final) the reference yields,
    for a run that ends on EOS between chunk boundaries (the final chunk then still holds the first sample)
    and a run that fills the cache; boost_first_chunk on and off."""
    cfg = synth.gpt_config()
    out = {}
    for name, seed, eos_gain, p, t, n, buckets, chunk, boost in [("e", 31, 2.5, 8, 14, 20, [(1, 128)], 10, True),
                                                                 ("f", 32, -8.0, 6, 9, 16, [(1, 64)], 8, False)]:
        w = synth.gpt_weights(cfg, seed=seed, eos_gain=eos_gain)
        m = build_ref_gpt(T2S, cfg, w, buckets)
        x, y, bert, _ = synth.synth_request(3, p, t, n, seed=seed)
        with torch.inference_mode():
            chunks = list(m.infer_stream(tt(x)[None], tt(y)[None], tt(bert)[None], top_k=1, stream_chunk=chunk,
                                         boost_first_chunk=boost, debug=False))
        out[name + "_cfg"] = np.array([seed, p, t, n, chunk, int(boost)])
        out[name + "_eos_gain"] = np.float32(eos_gain)
        out[name + "_cache"] = np.array(buckets)
        out[name + "_x"], out[name + "_y"] = x, y
        out[name + "_n"] = np.array(len(chunks))
        for i, (c, fin) in enumerate(chunks):
            out["%s_chunk%d" % (name, i)] = c[0, 0].numpy()
            out["%s_final%d" % (name, i)] = np.array(int(fin))
        print("stream", name, [(len(c[0, 0]), bool(f)) for c, f in chunks])
    np.savez_compressed(os.path.join(GOLD, "t2s_stream.npz"), **META, **out)


def gen_t2s_batched(T2S):
    cfg = synth.gpt_config()
    out = {}
    cases = [  # name, seed, eos_gain, [(p,t,n)...], buckets
        ("r", 1234, 6.0, [(8, 14, 20), (10, 20, 28), (6, 12, 16), (9, 25, 22), (7, 10, 30)], [(2, 128), (2, 160)]),
        ("s", 33, 1.0, [(8, 14, 20), (10, 20, 28), (6, 12, 16)], [(4, 96)]),
    ]
    for name, seed, eg, reqs, cache in cases:
        w = synth.gpt_weights(cfg, seed=seed, eos_gain=eg)
        m = build_ref_gpt(T2S, cfg, w, cache + [(1, cache[-1][1])])
        margins = with_margins(m)
        rs = [synth.synth_request(200 + i, p, t, n, seed=seed) for i, (p, t, n) in enumerate(reqs)]
        with torch.inference_mode():
            pred, orig = m.infer_batched([tt(r[0]) for r in rs], [tt(r[1]) for r in rs], [tt(r[2]) for r in rs], top_k=1)
        out[name + "_n"] = len(rs)
        out[name + "_reqs"] = np.array(reqs, np.int64)
        out[name + "_seed"] = seed
        out[name + "_eos_gain"] = eg
        out[name + "_cache"] = np.array(cache, np.int64)
        out[name + "_orig"] = orig.numpy()
        out[name + "_margin_min"] = min(margins)
        for i, p_ in enumerate(pred):
            out["%s_tok%d" % (name, i)] = p_.numpy()
        print("batched case", name, "orig", orig.tolist(), "lens", [len(p_) for p_ in pred], "min margin", min(margins))
    np.savez_compressed(os.path.join(GOLD, "t2s_batched.npz"), **META, **out)


def gen_sample(sample):
    out = {}
    g = np.random.default_rng(5)
    for i, (kw, prev) in enumerate([
        (dict(top_k=15, top_p=1.0, temperature=1.0, repetition_penalty=1.35), True),
        (dict(top_k=5, top_p=0.8, temperature=0.7, repetition_penalty=1.2), True),
        (dict(top_k=15, top_p=1.0, temperature=1.0, repetition_penalty=1.35), False),
        (dict(top_k=1, top_p=1.0, temperature=1.0, repetition_penalty=1.35), True),
        # many kept entries / the whole vocabulary: the device sampler finds the k-th largest by bisection from k = 65 on
        (dict(top_k=300, top_p=1.0, temperature=0.9, repetition_penalty=1.35), True),
        (dict(top_k=1025, top_p=1.0, temperature=1.0, repetition_penalty=1.0), False),
    ]):
        lg = (g.standard_normal((3, 1025)) * 3).astype(np.float32)
        lg[:, [280, 486]] = -np.inf
        pv = g.integers(0, 1025, size=(3, 40)) if prev else None
        torch.manual_seed(77 + i)
        idx, probs = sample(tt(lg.copy()), tt(pv) if prev else None, **kw)
        torch.manual_seed(77 + i)
        q = torch.empty_like(probs).exponential_(1)
        out.update({"c%d_logits" % i: lg, "c%d_q" % i: q.numpy(), "c%d_idx" % i: idx.numpy(),
                    "c%d_probs" % i: probs.numpy(), "c%d_kw" % i: np.array([kw["top_k"], kw["top_p"], kw["temperature"], kw["repetition_penalty"]], np.float64)})
        if prev:
            out["c%d_prev" % i] = pv
    np.savez_compressed(os.path.join(GOLD, "sample.npz"), **META, **out)
    print("sample ok")


def build_ref_sovits(Syn, hps, weights):
    s = Syn(hps["data"]["filter_length"] // 2 + 1, hps["train"]["segment_size"] // hps["data"]["hop_length"],
            n_speakers=hps["data"]["n_speakers"], **hps["model"])
    s.load_state_dict({k: tt(v) for k, v in weights.items()}, strict=False)
    s.dec.remove_weight_norm()
    s.eval()
    return s


def gen_vocoder(Syn):
    out = {}
    for ver, T, per_frame in [("v2Pro", 50, False), ("v2Pro", 55, True), ("v2ProPlus", 50, False), ("v2", 23, False),
                              ("v2Pro", 200, False), ("v2ProPlus", 55, True)]:
        hps = synth.sovits_hps(ver)
        w = synth.sovits_weights(hps, seed=1234)
        # weights were generated for the weight-norm-removed `dec`; load AFTER removing it
        s = Syn(hps["data"]["filter_length"] // 2 + 1, hps["train"]["segment_size"] // hps["data"]["hop_length"],
                n_speakers=hps["data"]["n_speakers"], **hps["model"])
        s.dec.remove_weight_norm()
        s.load_state_dict({k: tt(v) for k, v in w.items()}, strict=False)
        s.eval()
        gin = hps["model"]["gin_channels"]
        name = "%s_T%d_%s" % (ver, T, "pf" if per_frame else "c")
        z = synth.hashed_uniform(name + ".z", (1, 192, T)) * np.float32(1.7)
        mask = np.ones((1, 1, T), np.float32)
        if per_frame:
            ge = np.concatenate([np.repeat(synth.synth_ge(i, gin), n, axis=2) for i, n in ((0, 20), (1, T - 20))], axis=2)
        else:
            ge = synth.synth_ge(0, gin)
        with torch.inference_mode():
            zf = s.flow(tt(z), tt(mask), tt(ge))
            o = s.flow_dec(tt(z), tt(mask), tt(ge))
        out.update({name + "_z": z, name + "_ge": ge, name + "_flow": zf.numpy()})
        if T <= 64:
            out[name + "_o"] = o.numpy()[0, 0]
        else:   # long cases: every 5th sample + the fp64 sum of all of them (keeps the fixture small)
            out[name + "_o_sub"] = o.numpy()[0, 0, ::5].copy()
            out[name + "_o_sum"] = np.float64(o.numpy()[0, 0].astype(np.float64).sum())
        print("vocoder", name, "o std", o.std().item(), "max", o.abs().max().item())
    np.savez_compressed(os.path.join(GOLD, "vocoder.npz"), seed=1234, **META, **out)


def gen_decode(Syn):
    """SynthesizerTrn.decode end to end (quantizer lookup -> enc_p -> flow -> Generator), noise_scale=0:
    a single utterance, and the time-concatenated batch form with per-frame ge + slice_indices."""
    out = {}
    for ver in ("v2Pro", "v2"):
        hps = synth.sovits_hps(ver)
        w = synth.sovits_weights(hps, seed=1234)
        s = Syn(hps["data"]["filter_length"] // 2 + 1, hps["train"]["segment_size"] // hps["data"]["hop_length"],
                n_speakers=hps["data"]["n_speakers"], **hps["model"])
        s.dec.remove_weight_norm()
        s.load_state_dict({k: tt(v) for k, v in w.items()}, strict=False)
        s.eval()
        gin = hps["model"]["gin_channels"]
        N, P = 19, 13
        codes = synth.hashed_ints(ver + ".codes", N, 0, 1024)[None, None]
        text = synth.hashed_ints(ver + ".text", P, 1, 700)[None]
        ge = synth.synth_ge(0, gin)
        with torch.inference_mode():
            o, attn = s.decode(tt(codes), tt(text), tt(ge), noise_scale=0.0, speed=1, cuda_graph=False)
        out.update({ver + "_codes": codes, ver + "_text": text, ver + "_ge": ge, ver + "_o": o.numpy()[0, 0],
                    ver + "_attn": attn.numpy()})
        # two utterances concatenated along time (TTS.py:728-764)
        n1, n2, p1, p2 = 11, 8, 6, 7
        ge_cat = np.concatenate([np.repeat(synth.synth_ge(1, gin), n1, axis=2), np.repeat(synth.synth_ge(2, gin), n2, axis=2)], axis=2)
        pairs = np.array([[0, p1]] * (2 * n1) + [[p1, p1 + p2]] * (2 * n2), np.int64)
        with torch.inference_mode():
            ob, _ = s.decode(tt(codes), tt(text), tt(ge_cat), noise_scale=0.0, speed=1, cuda_graph=False, slice_indices=tt(pairs))
        out.update({ver + "_ge_cat": ge_cat, ver + "_pairs": pairs, ver + "_ob": ob.numpy()[0, 0]})
        print("decode", ver, o.shape, "std", o.std().item())
    np.savez_compressed(os.path.join(GOLD, "decode.npz"), seed=1234, **META, **out)


ALIGN_CASES = [  # (seed, heads, frames, phonemes, lead, tail)
    (100, 4, 60, 17, 0, 0), (101, 4, 60, 17, 5, 6), (102, 4, 200, 40, 3, 20), (103, 4, 500, 120, 0, 10),
    (104, 2, 64, 2, 0, 4), (105, 4, 33, 3, 2, 2), (106, 4, 300, 300, 4, 9), (107, 4, 40, 70, 0, 0),
    (108, 4, 1, 5, 0, 0), (109, 4, 700, 90, 10, 30), (110, 4, 150, 1100, 2, 5), (111, 8, 90, 2500, 0, 3),
]


def _word2ph(n_ph, seed):
    """words of 1-4 phonemes covering exactly n_ph phonemes (the last word is a pause mark when it is 1 long)"""
    rng = np.random.default_rng(seed)
    words, phs, left, k = [], [], n_ph, 0
    while left > 0:
        c = int(min(left, rng.integers(1, 5)))
        words.append("w%d" % k)
        phs.append(c)
        left -= c
        k += 1
    if phs[-1] == 1:
        words[-1] = "."
    return {"word": words, "ph": phs}


def gen_subtitles():
    """Subtitle alignment + host bookkeeping: TTS._viterbi_monotonic / _is_normal_assign / _get_subtitles /
    _find_subtitles / _cat_subtitles and TextProcessor.sub2text_index, run from the reference's own source.
    TTS.py cannot be imported here (av / torchaudio are absent), so the function definitions are compiled from
    the reference file at generation time (ref_harness.reference_functions); only their outputs are stored.
    The attention inputs are regenerated from a seed (synth.synth_attn); the fixture keeps a checksum."""
    import bisect, copy, json, re
    tf = reference_functions("gsv_tts/TTS.py", ["_viterbi_monotonic", "_is_normal_assign", "_get_subtitles", "_find_subtitles",
                                               "_cat_subtitles"], {"torch": torch})

    class Self:            # the attributes those methods read
        sovits_hz = 50
    out, host = {}, {"get_subtitles": [], "is_normal_assign": [], "find_subtitles": [], "cat_subtitles": [], "sub2text_index": []}
    for (seed, H, T, P, lead, tail) in ALIGN_CASES:
        a = synth.synth_attn(seed, H, T, P, lead, tail)
        assign = tf["_viterbi_monotonic"](Self(), tt(a))
        out["assign_%d" % seed] = assign.numpy()
        out["check_%d" % seed] = np.float64(a.astype(np.float64).sum())
        host["is_normal_assign"].append({"seed": seed, "want": bool(tf["_is_normal_assign"](Self(), assign))})
        if T > 1 and P <= 300:
            for speed, last_end in ((1.0, 0), (1.3, 1.7)):
                for extra in (0, 3):   # word list longer than the aligned phonemes -> the early-break branch
                    w2p = _word2ph(P + extra, seed)
                    subs = tf["_get_subtitles"](Self(), w2p, assign, speed, last_end_s=last_end)
                    host["get_subtitles"].append({"seed": seed, "word2ph": w2p, "speed": speed, "last_end_s": last_end, "want": subs})
    out["cases"] = np.array(ALIGN_CASES, np.int64)
    for assign in ([-1, -1, -1], [0, 1, 2, 3], [0, 0, 1, 2, 2, 3], [-1, 0, 0, 1, 1], [5], [2, 2, 3, 3, 4, 5, 6, 6]):
        host["is_normal_assign"].append({"assign": assign, "want": bool(tf["_is_normal_assign"](Self(), torch.tensor(assign)))})
    mk = lambda words, t0=0.0: [{"text": w, "start_s": t0 + 0.25 * i, "end_s": t0 + 0.25 * (i + 1)} for i, w in enumerate(words)]
    subs = mk(["a", "b", ".", "c", "d", "e", "!", "a", "b", "."])
    for w2p, last_i in (({"word": ["a", "b", "."]}, 0), ({"word": ["c", "d", "e", "!"]}, 3), ({"word": ["a", "b", "."]}, 3),
                        ({"word": ["x", "y"]}, 2), ({"word": ["a"] * 12}, 0), ({"word": ["b", "."]}, 9)):
        host["find_subtitles"].append({"subtitles": subs, "word2ph": w2p, "last_i": last_i,
                                       "want": int(tf["_find_subtitles"](Self(), subs, w2p, last_i))})
    lists = [mk(["a", "b"], 0.5), mk(["c"], 3.0), mk(["d", "e", "f"], 0.125)]
    host["cat_subtitles"].append({"lists": copy.deepcopy(lists), "want": tf["_cat_subtitles"](Self(), *copy.deepcopy(lists))})
    pf = reference_functions("gsv_tts/TextProcessor.py", ["split_text", "LIS_mapping", "linear_interpolate", "sub2text_index"],
                             {"re": re, "bisect": bisect})
    texts = [
        ("hello world, this is a test.", "Hello world, this is a test.", ["hello", "world", ",", "this", "is", "a", "test", "."]),
        ("it costs five dollars today.", "It costs $5 today.", ["it", "costs", "five", "dollars", "today", "."]),
        ("\u4eca\u5929\u5929\u6c14\u5f88\u597d.", "\u4eca\u5929\u5929\u6c14\u5f88\u597d\uff01", ["\u4eca\u5929", "\u5929\u6c14", "\u5f88", "\u597d", "."]),
        ("a a a b a.", "a b a a a.", ["a", "a", "a", "b", "a", "."]),
        ("one two three.", "1 2 3", ["one", "two", "three", "."]),
        ("x y z.", "x y z.", ["x", "q", "z", "."]),
    ]
    for norm, orig, words in texts:
        subs = mk(words)
        host["sub2text_index"].append({"norm_text": norm, "orig_text": orig, "subtitles": copy.deepcopy(subs),
                                       "want": pf["sub2text_index"](copy.deepcopy(subs), norm, orig)})
    for cand in ([[0], [1], [2]], [[2, 5], [0, 3], [1, 4], []], [[], []], [[3, 1], [2], [3, 4], [0, 5]], [[1, 2, 3]] * 4):
        host.setdefault("lis_mapping", []).append({"candidates": cand, "want": pf["LIS_mapping"]([list(c) for c in cand])})
    for idx in ([-1, -1, 4, -1, -1, 9, -1], [-1] * 3, [0, -1, -1, -1, 7], [-1, -1, -1, 2], [5, -1, -1]):
        host.setdefault("linear_interpolate", []).append({"indices": idx, "want": pf["linear_interpolate"](list(idx))})
    np.savez_compressed(os.path.join(GOLD, "align.npz"), **META, **out)
    with open(os.path.join(GOLD, "subtitles.json"), "w") as f:
        json.dump(host, f, ensure_ascii=True, indent=1)
    print("subtitles:", len(ALIGN_CASES), "alignments,", {k: len(v) for k, v in host.items()})


REF_CASES = [("v2Pro", 0, 32000 * 3 + 123, 151), ("v2", 1, 40000, 64), ("v2ProPlus", 2, 2048, 3)]


def gen_refaudio(Syn):
    """Reference-audio path: SynthesizerTrn.get_ge / extract_latent of the imported reference, and the Spectrogram
    of TTS._get_spec.  torchaudio (a third-party dependency, absent here and unpinned by the reference) is not
    importable; its Spectrogram(power=1) is |torch.stft(...)| with the arguments TTS.py:1591-1604 passes, which is
    what runs here.  Inputs and weights are regenerated from seeds; the spectrogram is stored subsampled."""
    out = {}
    for ver, i, n_samples, n_ssl in REF_CASES:
        hps = synth.sovits_hps(ver)
        m = Syn(1025, 32, n_speakers=300, **hps["model"]).eval()
        sd = {k: tt(v) for k, v in synth.ref_audio_weights(hps, 1234).items()}
        sd["quantizer.vq.layers.0._codebook.inited"] = torch.ones(1)   # a trained checkpoint: no k-means re-init
        assert not m.load_state_dict(sd, strict=False).unexpected_keys
        a = synth.synth_audio(i, n_samples)
        spec = torch.stft(tt(a), 2048, 640, 2048, window=torch.hann_window(2048), center=True, pad_mode="reflect",
                          normalized=False, onesided=True, return_complex=True).abs()
        sv = synth.synth_sv_emb(i) if ver != "v2" else None
        ssl = synth.synth_ssl(i, n_ssl)
        with torch.inference_mode():
            ge = m.get_ge(spec[None], tt(sv) if sv is not None else None)
            codes = m.extract_latent(tt(ssl))
        out[ver + "_spec_sub"] = spec.numpy()[::8, ::4].copy()
        out[ver + "_spec_sum"] = np.float64(spec.numpy().astype(np.float64).sum())
        out[ver + "_ge"] = ge.numpy()
        out[ver + "_codes"] = codes.numpy()
        print("refaudio", ver, tuple(spec.shape), "ge std %.3f" % ge.std().item(), "codes", tuple(codes.shape))
    np.savez_compressed(os.path.join(GOLD, "refaudio.npz"), seed=1234, **META, **out)


from gen_golden_inputs import FACADE_AUDIO, FACADE_LENGTHS, FACADE_SPLITS, SOLA_CASES, facade_audio, sola_case  # noqa: E402


def gen_facade():
    """The facade's own arithmetic, executed from the reference's source (TTS.py cannot be imported here):
      TTS._find_head_threshold_offsets / _find_tail_threshold_offsets  TTS.py:1629-1662  (functions)
      the sort + both-ends interleave of infer_batched                  TTS.py:705-720   (inline statements)
      the split / trim loop of infer_batched                            TTS.py:806-816   (inline statements)
    Inputs are regenerated from seeds (facade_audio, FACADE_LENGTHS); only outputs are stored."""
    tf = reference_functions("gsv_tts/TTS.py", ["_find_head_threshold_offsets", "_find_tail_threshold_offsets"], {"torch": torch})
    out = {}
    head, tail = [], []
    for case in FACADE_AUDIO:
        a = tt(facade_audio(*case))
        head.append(tf["_find_head_threshold_offsets"](None, a))
        tail.append(tf["_find_tail_threshold_offsets"](None, a))
    out["head"], out["tail"] = np.array(head, np.int64), np.array(tail, np.int64)
    sort_code = reference_statements("gsv_tts/TTS.py", "infer_batched", "semantic_lengths = torch.tensor(", "semantic_lengths = semantic_lengths[idx_map]")

    class Cfg:
        device = torch.device("cpu")

    class Self:
        tts_config = Cfg()
        _find_head_threshold_offsets = tf["_find_head_threshold_offsets"]
        _find_tail_threshold_offsets = tf["_find_tail_threshold_offsets"]
    for k, lens in enumerate(FACADE_LENGTHS):
        ns = {"torch": torch, "self": Self(), "pred_semantic": [torch.zeros(l, dtype=torch.int64) for l in lens],
              "semantic_orig_idx": torch.arange(100, 100 + len(lens))}
        exec(sort_code, ns)
        out["order_%d" % k] = ns["idx_map"].numpy()
        out["orig_%d" % k] = ns["semantic_orig_idx"].numpy()
    split_code = reference_statements("gsv_tts/TTS.py", "infer_batched", "last_actual_len = 0", "for j in range(len(semantic_list))")

    class VQ:
        samples_per_frame = 640
    for k, (lens, speed) in enumerate(FACADE_SPLITS):
        total = int(sum(lens) * 2 * 640 / speed) + 1
        audio = tt(facade_audio(20 + k, total, 0, 0, 0.4))
        pos = 0
        for i, l in enumerate(lens):       # a quiet gap at the head of every utterance, so that the trims differ
            audio[int(pos): int(pos) + 700 * (i + 1)] = 0
            pos += l * 2 * 640 / speed
        ns = {"torch": torch, "np": np, "self": Self(), "semantic_list": [None] * len(lens), "curr_lengths": torch.tensor(lens),
              "vq_model": VQ(), "speed": speed, "audio_batch": audio, "generated_audios": []}
        exec(split_code, ns)
        out["split_%d_lens" % k] = np.array(lens, np.int64)
        out["split_%d_speed" % k] = np.float64(speed)
        out["split_%d_sizes" % k] = np.array([len(a) for a in ns["generated_audios"]], np.int64)
        out["split_%d_sums" % k] = np.array([float(np.abs(a).astype(np.float64).sum()) for a in ns["generated_audios"]])
    np.savez_compressed(os.path.join(GOLD, "facade.npz"), **META, **out)
    print("facade: head", head, "tail", tail)


def gen_sola():
    """TTS._sola_algorithm (TTS.py:1612-1627), executed from the reference's source on the seeded cases of
    gen_golden_inputs.SOLA_CASES: the chosen offset and the spliced chunk.  Only outputs are stored."""
    import torch.nn.functional as F
    fn = reference_functions("gsv_tts/TTS.py", ["_sola_algorithm"], {"torch": torch, "F": F})["_sola_algorithm"]

    class Cfg:
        device = torch.device("cpu")
        dtype = torch.float32

    class Self:
        tts_config = Cfg()
    out = {}
    for k, case in enumerate(SOLA_CASES):
        f1, f2 = sola_case(*case)
        real, off = fn(Self(), tt(f1)[None, None], tt(f2)[None, None], case[2], case[3])
        out["offset_%d" % k] = np.int64(int(off.item()))
        out["out_%d" % k] = real[0, 0].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "sola.npz"), **META, **out)
    print("sola offsets:", [int(out["offset_%d" % k]) for k in range(len(SOLA_CASES))])


if __name__ == "__main__":
    tqdm.tqdm.__init__ = functools.partialmethod(tqdm.tqdm.__init__, disable=True)
    torch.manual_seed(0)
    T2S, sample, Syn = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["layers", "infer", "batched", "stream", "sample", "vocoder", "decode", "subtitles", "refaudio", "facade", "sola"]
    if "layers" in which: gen_t2s_layers(T2S)
    if "infer" in which: gen_t2s_infer(T2S)
    if "batched" in which: gen_t2s_batched(T2S)
    if "stream" in which: gen_t2s_stream(T2S)
    if "sample" in which: gen_sample(sample)
    if "vocoder" in which: gen_vocoder(Syn)
    if "decode" in which: gen_decode(Syn)
    if "subtitles" in which: gen_subtitles()
    if "refaudio" in which: gen_refaudio(Syn)
    if "facade" in which: gen_facade()
    if "sola" in which: gen_sola()
