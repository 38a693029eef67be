"""Deterministic synthetic weights and inputs for the GPT-SoVITS hot path.

There are no checkpoints in this environment, so
parity fixtures, `smoke()` and `bench.py` all run on *seeded* weights of the
real architecture.  The generator is a pure integer hash (FNV-1a of the tensor
name -> splitmix64 per element -> 24-bit uniform), evaluated with numpy uint64
arithmetic, so the same name/shape/seed yields bit-identical float32 values on
any box -- the golden vectors in tests/golden/ were produced by feeding exactly
these tensors to the imported reference (oracle/gen_golden.py).

Tensor names and shapes follow the reference's state dicts after its key remap
(reference gsv_tts/Loader.py:130-154 for GPT; SoVITS names as saved by
`SynthesizerTrn.state_dict()` with `dec.remove_weight_norm()` applied,
Loader.py:94-95), so the same dicts load into the reference modules and into
this package's loaders.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hashed_uniform(name: str, shape, seed: int = 1234) -> np.ndarray:
    """float32 array of `shape`, iid uniform in (-1, 1), a pure function of (name, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    bits = _splitmix64(idx) >> np.uint64(40)  # top 24 bits
    u = (bits.astype(np.float64) + 0.5) / float(1 << 24)
    return (2.0 * u - 1.0).astype(np.float32).reshape(shape)


def hashed_ints(name: str, n: int, lo: int, hi: int, seed: int = 1234) -> np.ndarray:
    """int64[n] uniform in [lo, hi), pure function of (name, seed)."""
    base = np.uint64((_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    bits = _splitmix64(idx) >> np.uint64(11)
    return (lo + (bits % np.uint64(hi - lo)).astype(np.int64)).astype(np.int64)


_SQRT3 = math.sqrt(3.0)


def _std(name, shape, std, seed, mean=0.0):
    return (mean + hashed_uniform(name, shape, seed) * np.float32(std * _SQRT3)).astype(np.float32)


# --------------------------------------------------------------------------------------
# GPT (Text2SemanticDecoder) -- reference gsv_tts/GPT_SoVITS/GPT/t2s_model.py:158-206
# --------------------------------------------------------------------------------------

def gpt_config(n_layer: int = 24, hidden: int = 512, head: int = 16, vocab: int = 1025,
               phoneme_vocab: int = 732) -> dict:
    """[upstream] standard s1 config (SURVEY.md section 8 preamble)."""
    return {"model": {"hidden_dim": hidden, "embedding_dim": hidden, "head": head,
                      "n_layer": n_layer, "vocab_size": vocab,
                      "phoneme_vocab_size": phoneme_vocab, "dropout": 0.0, "EOS": vocab - 1}}


def gpt_spec(config: dict) -> "OrderedDict[str, tuple]":
    m = config["model"]
    D, V, P, L = m["hidden_dim"], m["vocab_size"], m["phoneme_vocab_size"], m["n_layer"]
    s = OrderedDict()
    s["bert_proj.weight"] = (D, 1024)
    s["bert_proj.bias"] = (D,)
    s["ar_text_embedding.word_embeddings.weight"] = (P, D)
    s["ar_text_position.alpha"] = (1,)
    s["ar_audio_embedding.word_embeddings.weight"] = (V, D)
    s["ar_audio_position.alpha"] = (1,)
    s["ar_predict_layer.weight"] = (V, D)
    for i in range(L):
        p = "t2s_transformer.blocks.%d." % i
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "qkv.weight"] = (3 * D, D)
        s[p + "qkv.bias"] = (3 * D,)
        s[p + "out_proj.weight"] = (D, D)
        s[p + "out_proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.0.weight"] = (4 * D, D)
        s[p + "mlp.0.bias"] = (4 * D,)
        s[p + "mlp.2.weight"] = (D, 4 * D)
        s[p + "mlp.2.bias"] = (D,)
    return s


def gpt_weights(config: dict, seed: int = 1234, logit_gain: float = 6.0,
                eos_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Seeded GPT weights.  `logit_gain` widens top-1/top-2 margins so greedy tokens are
    robust to fp32 summation-order noise (SURVEY.md section 7 'hard parts'); `eos_gain`
    scales the EOS row of the predict layer (0 -> EOS never wins -> fixed-length runs,
    >1 -> EOS appears early, for stop-logic tests)."""
    m = config["model"]
    D = m["hidden_dim"]
    out = OrderedDict()
    for name, shape in gpt_spec(config).items():
        if name.endswith("alpha"):
            v = np.array([0.9 if "text" in name else 1.1], dtype=np.float32)
        elif name.endswith("word_embeddings.weight"):
            v = _std(name, shape, 1.0, seed)
        elif name == "bert_proj.weight":
            v = _std(name, shape, 1.0 / math.sqrt(1024), seed)
        elif name == "ar_predict_layer.weight":
            v = _std(name, shape, logit_gain / math.sqrt(D), seed)
            v[m["EOS"]] *= np.float32(eos_gain)
        elif name.endswith("qkv.weight"):
            v = _std(name, shape, 1.5 / math.sqrt(D), seed)
        elif name.endswith("mlp.0.weight"):
            v = _std(name, shape, math.sqrt(2.0) / math.sqrt(D), seed)
        elif name.endswith("mlp.2.weight"):
            v = _std(name, shape, 0.15 / math.sqrt(4 * D), seed)
        elif name.endswith("out_proj.weight"):
            # small residual branches: a deep random post-LN stack with O(1) branches collapses
            # to an input-independent fixed point (every greedy token identical)
            v = _std(name, shape, 0.15 / math.sqrt(D), seed)
        elif ".norm" in name and name.endswith("weight"):
            v = _std(name, shape, 0.1, seed, mean=1.0)
        elif ".norm" in name and name.endswith("bias"):
            v = _std(name, shape, 0.1, seed)
        elif name.endswith("bias"):
            v = _std(name, shape, 0.05, seed)
        else:  # pragma: no cover
            raise KeyError(name)
        out[name] = v
    return out


# --------------------------------------------------------------------------------------
# SoVITS (SynthesizerTrn) -- reference gsv_tts/GPT_SoVITS/SoVITS/models.py:235-320
# --------------------------------------------------------------------------------------

def sovits_hps(version: str = "v2Pro") -> dict:
    """[upstream] s2 hyper-parameters (SURVEY.md section 8 preamble)."""
    assert version in ("v2", "v2Pro", "v2ProPlus")
    return {
        "data": {"filter_length": 2048, "hop_length": 640, "n_speakers": 300,
                 "sampling_rate": 32000},
        "train": {"segment_size": 20480},
        "model": {
            "inter_channels": 192, "hidden_channels": 192, "filter_channels": 768,
            "n_heads": 2, "n_layers": 6, "kernel_size": 3, "p_dropout": 0.0,
            "resblock": "1", "resblock_kernel_sizes": [3, 7, 11],
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
            "upsample_rates": [10, 8, 2, 2, 2],
            "upsample_initial_channel": 768 if version == "v2ProPlus" else 512,
            "upsample_kernel_sizes": [16, 16, 8, 2, 2],
            "gin_channels": 512 if version == "v2" else 1024,
            "semantic_frame_rate": "25hz", "version": version,
        },
    }


def _encoder_spec(s, p, hidden, filt, n_heads, n_layers, ksz, window=4):
    kc = hidden // n_heads
    for i in range(n_layers):
        a = "%sattn_layers.%d." % (p, i)
        s[a + "emb_rel_k"] = (1, 2 * window + 1, kc)
        s[a + "emb_rel_v"] = (1, 2 * window + 1, kc)
        for c in ("conv_q", "conv_k", "conv_v", "conv_o"):
            s[a + c + ".weight"] = (hidden, hidden, 1)
            s[a + c + ".bias"] = (hidden,)
        s["%snorm_layers_1.%d.gamma" % (p, i)] = (hidden,)
        s["%snorm_layers_1.%d.beta" % (p, i)] = (hidden,)
        f = "%sffn_layers.%d." % (p, i)
        s[f + "conv_1.weight"] = (filt, hidden, ksz)
        s[f + "conv_1.bias"] = (filt,)
        s[f + "conv_2.weight"] = (hidden, filt, ksz)
        s[f + "conv_2.bias"] = (hidden,)
        s["%snorm_layers_2.%d.gamma" % (p, i)] = (hidden,)
        s["%snorm_layers_2.%d.beta" % (p, i)] = (hidden,)


def sovits_spec(hps: dict, n_symbols: int = 732, hot_path_only: bool = False) -> "OrderedDict[str, tuple]":
    """Names/shapes of the tensors `decode()` touches (enc_p, quantizer codebook, flow, dec,
    ge_to512).  `ref_enc`/`ssl_proj`/`sv_emb`/`prelu` (reference-audio path, SURVEY section
    8(f) rank 3) are not generated: the reference loads with strict=False (Loader.py:94)."""
    m = hps["model"]
    H, F, NH, NL, K = m["hidden_channels"], m["filter_channels"], m["n_heads"], m["n_layers"], m["kernel_size"]
    inter, gin, C0 = m["inter_channels"], m["gin_channels"], m["upsample_initial_channel"]
    s = OrderedDict()
    if not hot_path_only:
        s["enc_p.ssl_proj.weight"] = (H, 768, 1)
        s["enc_p.ssl_proj.bias"] = (H,)
        _encoder_spec(s, "enc_p.encoder_ssl.", H, F, NH, NL // 2, K)
        _encoder_spec(s, "enc_p.encoder_text.", H, F, NH, NL, K)
        s["enc_p.text_embedding.weight"] = (n_symbols, H)
        for c in ("conv_q", "conv_k", "conv_v", "conv_o"):
            s["enc_p.mrte.cross_attention.%s.weight" % c] = (512, 512, 1)
            s["enc_p.mrte.cross_attention.%s.bias" % c] = (512,)
        s["enc_p.mrte.c_pre.weight"] = (512, H, 1)
        s["enc_p.mrte.c_pre.bias"] = (512,)
        s["enc_p.mrte.text_pre.weight"] = (512, H, 1)
        s["enc_p.mrte.text_pre.bias"] = (512,)
        s["enc_p.mrte.c_post.weight"] = (H, 512, 1)
        s["enc_p.mrte.c_post.bias"] = (H,)
        _encoder_spec(s, "enc_p.encoder2.", H, F, NH, NL // 2, K)
        s["enc_p.proj.weight"] = (2 * inter, H, 1)
        s["enc_p.proj.bias"] = (2 * inter,)
        s["quantizer.vq.layers.0._codebook.embed"] = (1024, 768)
        if m["version"] in ("v2Pro", "v2ProPlus"):
            s["ge_to512.weight"] = (512, gin)
            s["ge_to512.bias"] = (512,)
    # Generator (weight-norm already removed, Loader.py:95)
    s["dec.conv_pre.weight"] = (C0, inter, 7)
    s["dec.conv_pre.bias"] = (C0,)
    s["dec.cond.weight"] = (C0, gin, 1)
    s["dec.cond.bias"] = (C0,)
    ch = C0
    for i, (u, k) in enumerate(zip(m["upsample_rates"], m["upsample_kernel_sizes"])):
        s["dec.ups.%d.weight" % i] = (ch, ch // 2, k)  # ConvTranspose1d layout [Cin, Cout, k]
        s["dec.ups.%d.bias" % i] = (ch // 2,)
        ch //= 2
        for j, rk in enumerate(m["resblock_kernel_sizes"]):
            r = "dec.resblocks.%d." % (i * len(m["resblock_kernel_sizes"]) + j)
            for c in ("convs1", "convs2"):
                for d in range(3):
                    s["%s%s.%d.weight" % (r, c, d)] = (ch, ch, rk)
                    s["%s%s.%d.bias" % (r, c, d)] = (ch,)
    s["dec.conv_post.weight"] = (1, ch, 7)
    # Flow: weight-norm (g, v) parameters stay live at inference
    for fl in range(0, 8, 2):
        p = "flow.flows.%d." % fl
        s[p + "pre.weight"] = (H, inter // 2, 1)
        s[p + "pre.bias"] = (H,)
        for l in range(4):
            s["%senc.in_layers.%d.bias" % (p, l)] = (2 * H,)
            s["%senc.in_layers.%d.weight_g" % (p, l)] = (2 * H, 1, 1)
            s["%senc.in_layers.%d.weight_v" % (p, l)] = (2 * H, H, 5)
            rs = 2 * H if l < 3 else H
            s["%senc.res_skip_layers.%d.bias" % (p, l)] = (rs,)
            s["%senc.res_skip_layers.%d.weight_g" % (p, l)] = (rs, 1, 1)
            s["%senc.res_skip_layers.%d.weight_v" % (p, l)] = (rs, H, 1)
        s[p + "enc.cond_layer.bias"] = (8 * H,)
        s[p + "enc.cond_layer.weight_g"] = (8 * H, 1, 1)
        s[p + "enc.cond_layer.weight_v"] = (8 * H, gin, 1)
        s[p + "post.weight"] = (inter // 2, H, 1)
        s[p + "post.bias"] = (inter // 2,)
    return s


def sovits_weights(hps: dict, seed: int = 1234, hot_path_only: bool = False) -> "OrderedDict[str, np.ndarray]":
    out = OrderedDict()
    spec = sovits_spec(hps, hot_path_only=hot_path_only)
    for name, shape in spec.items():
        if name.endswith("weight_g"):
            continue  # derived from weight_v below
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if name.endswith("weight_v"):
            gain = 0.5 if "cond_layer" in name else 1.0
            v = _std(name, shape, gain / math.sqrt(fan_in), seed)
            gname = name[:-1] + "g"
            nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True)).astype(np.float32)
            out[gname] = (nrm * (1.0 + 0.1 * hashed_uniform(gname, spec[gname], seed))).astype(np.float32)
        elif name.endswith("emb_rel_k") or name.endswith("emb_rel_v"):
            v = _std(name, shape, shape[-1] ** -0.5, seed)
        elif name.endswith("gamma"):
            v = _std(name, shape, 0.1, seed, mean=1.0)
        elif name.endswith("beta"):
            v = _std(name, shape, 0.1, seed)
        elif name == "enc_p.text_embedding.weight":
            v = _std(name, shape, shape[1] ** -0.5, seed)
        elif name.endswith("_codebook.embed"):
            v = _std(name, shape, 1.0, seed)
        elif name.startswith("dec.ups.") and name.endswith("weight"):
            cin, cout, k = shape
            i = int(name.split(".")[2])
            u = hps["model"]["upsample_rates"][i]
            v = _std(name, shape, 1.3 / math.sqrt(cin * max(1.0, k / u)), seed)
        elif ".convs2." in name and name.endswith("weight"):
            v = _std(name, shape, 0.6 / math.sqrt(fan_in), seed)  # modest residual branch
        elif ".convs1." in name and name.endswith("weight"):
            v = _std(name, shape, 1.3 / math.sqrt(fan_in), seed)
        elif name == "dec.conv_post.weight":
            v = _std(name, shape, 1.0 / math.sqrt(fan_in), seed)
        elif name == "dec.cond.weight":
            v = _std(name, shape, 0.3 / math.sqrt(fan_in), seed)
        elif name.endswith("post.weight") and name.startswith("flow."):
            v = _std(name, shape, 0.5 / math.sqrt(fan_in), seed)  # zero-init in a fresh module
        elif name.endswith("weight"):
            v = _std(name, shape, 1.0 / math.sqrt(fan_in), seed)
        elif name.endswith("bias"):
            v = _std(name, shape, 0.05, seed)
        else:  # pragma: no cover
            raise KeyError(name)
        out[name] = v
    # keep spec order
    return OrderedDict((k, out[k]) for k in spec)


# --------------------------------------------------------------------------------------
# Synthetic requests (SURVEY.md section 8(d) / BASELINE.md section 3)
# --------------------------------------------------------------------------------------

def synth_request(i: int, n_prompt_ph: int = 40, n_text_ph: int = 60, n_prompt_tok: int = 100,
                  seed: int = 1234, bert: str = "zeros", phoneme_vocab: int = 732):
    """One utterance's GPT inputs: (x int64[Lx], y int64[Ly], bert float32[Lx,1024], phones2)."""
    lx = n_prompt_ph + n_text_ph
    x = hashed_ints("req%d.x" % i, lx, 1, min(700, phoneme_vocab), seed)
    y = hashed_ints("req%d.y" % i, n_prompt_tok, 0, 1024, seed)
    if bert == "zeros":  # what the reference feeds for ja/en text (TextProcessor.py:100)
        b = np.zeros((lx, 1024), dtype=np.float32)
    else:
        b = hashed_uniform("req%d.bert" % i, (lx, 1024), seed) * np.float32(0.5)
    return x, y, b, x[n_prompt_ph:].copy()


def synth_ge(i: int, gin: int = 1024, seed: int = 1234) -> np.ndarray:
    """Reference-speaker embedding ge [1, gin, 1] (what get_ge would have produced)."""
    return (hashed_uniform("spk%d.ge" % i, (1, gin, 1), seed) * np.float32(_SQRT3)).astype(np.float32)


def mixed_lengths(n: int, seed: int = 1234):
    """Config-3 style mixed-length request set: (n_text_ph, n_prompt_tok) per request."""
    a = hashed_ints("mixed.text", n, 20, 121, seed)
    b = hashed_ints("mixed.prompt", n, 75, 151, seed)
    return [(int(p), int(q)) for p, q in zip(a, b)]


def mixed_new_tokens(n: int, seed: int = 1234):
    """generated-length targets of the mixed-length request set: N ~ U[50, 400]"""
    return [int(v) for v in hashed_ints("mixed.new", n, 50, 401, seed)]


def synth_attn(seed: int, heads: int, frames: int, phonemes: int, lead: int = 0, tail: int = 0,
               noise: float = 0.2) -> np.ndarray:
    """Synthetic MRTE cross-attention [heads, frames, phonemes] for the subtitle-alignment tests: a monotone
    ridge from phoneme 0 to phonemes-1 with per-head jitter and uniform noise.  `lead` frames before the ridge
    peak away from phoneme 0 (-> -1 in the alignment), `tail` frames peak at the last phoneme on every head
    (-> the fixed row of TTS.py:1757-1761).  Only IEEE +, *, / on float32, so it is bit-reproducible."""
    rng = np.random.default_rng(seed)
    span = max(1, frames - lead - tail)
    t = np.arange(frames)
    centre = np.clip((t - lead) * (phonemes - 1) // max(1, span - 1), 0, phonemes - 1)
    centre = np.where(t < lead, min(2, phonemes - 1), centre)
    centre = np.where(t >= frames - tail, phonemes - 1, centre)
    jit = rng.integers(-1, 2, size=(heads, frames))
    jit[:, t < lead] = 0
    jit[:, t >= frames - tail] = 0
    c = np.clip(centre[None, :] + jit, 0, phonemes - 1)
    if lead > 0:
        c[:, lead] = 0                      # the ridge starts on phoneme 0 for every head
    d = (np.arange(phonemes)[None, None, :] - c[:, :, None]).astype(np.float32)
    u = rng.random((heads, frames, phonemes), dtype=np.float32)
    return (np.float32(1) / (np.float32(1) + d * d) + np.float32(noise) * u).astype(np.float32)


# --------------------------------------------------------------------------------------
# Reference-audio path: ref_enc / sv_emb / prelu / ssl_proj tensors + inputs
# --------------------------------------------------------------------------------------

def ref_audio_spec(hps: dict) -> "OrderedDict[str, tuple]":
    """Names/shapes as the reference's SynthesizerTrn.state_dict() has them (SoVITS/models.py:305-318)."""
    gin = hps["model"]["gin_channels"]
    s = OrderedDict()
    s["ref_enc.spectral.0.fc.weight"] = (128, 704)
    s["ref_enc.spectral.0.fc.bias"] = (128,)
    s["ref_enc.spectral.3.fc.weight"] = (128, 128)
    s["ref_enc.spectral.3.fc.bias"] = (128,)
    for i in (0, 1):
        s["ref_enc.temporal.%d.conv1.conv.weight" % i] = (256, 128, 5)
        s["ref_enc.temporal.%d.conv1.conv.bias" % i] = (256,)
    for n in ("w_qs", "w_ks", "w_vs", "fc"):
        s["ref_enc.slf_attn.%s.weight" % n] = (128, 128)
        s["ref_enc.slf_attn.%s.bias" % n] = (128,)
    s["ref_enc.fc.fc.weight"] = (gin, 128)
    s["ref_enc.fc.fc.bias"] = (gin,)
    s["ssl_proj.weight"] = (768, 768, 2)
    s["ssl_proj.bias"] = (768,)
    s["quantizer.vq.layers.0._codebook.embed"] = (1024, 768)
    if hps["model"]["version"] in ("v2Pro", "v2ProPlus"):
        s["sv_emb.weight"] = (gin, 20480)
        s["sv_emb.bias"] = (gin,)
        s["prelu.weight"] = (gin,)
    return s


def ref_audio_weights(hps: dict, seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """Seeded tensors for ref_audio_spec; the codebook is the same tensor sovits_weights() generates.  The first
    linear is scaled for |STFT| inputs of O(30), the attention projections for logits with a spread of a few units."""
    out = OrderedDict()
    for name, shape in ref_audio_spec(hps).items():
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if name.endswith("_codebook.embed"):
            v = _std(name, shape, 1.0, seed)
        elif name == "ref_enc.spectral.0.fc.weight":
            v = _std(name, shape, 0.05 / math.sqrt(fan_in), seed)
        elif name in ("ref_enc.slf_attn.w_qs.weight", "ref_enc.slf_attn.w_ks.weight"):
            v = _std(name, shape, 2.5 / math.sqrt(fan_in), seed)
        elif name == "prelu.weight":
            v = _std(name, shape, 0.05, seed, mean=0.25)
        elif name.endswith("weight"):
            v = _std(name, shape, 1.0 / math.sqrt(fan_in), seed)
        else:
            v = _std(name, shape, 0.05, seed)
        out[name] = v
    return out


def synth_audio(i: int, n_samples: int, seed: int = 1234) -> np.ndarray:
    """Mono waveform in [-1, 1]: a few drifting partials + noise (only IEEE ops of float64 -> float32)."""
    t = np.arange(n_samples, dtype=np.float64) / 32000.0
    f = 110.0 + 30.0 * (hashed_uniform("aud%d.f" % i, (6,), seed).astype(np.float64) + 1.0)
    x = np.zeros(n_samples, np.float64)
    for k in range(6):
        x += (0.5 / (k + 1)) * np.sin(2 * np.pi * f[k] * (k + 1) * t * (1.0 + 0.05 * t))
    x += 0.05 * hashed_uniform("aud%d.n" % i, (n_samples,), seed).astype(np.float64)
    return (0.3 * x).astype(np.float32)


def synth_sv_emb(i: int, seed: int = 1234) -> np.ndarray:
    """ERes2Net speaker-verification embedding stand-in, [1, 20480]."""
    return (hashed_uniform("spk%d.sv" % i, (1, 20480), seed) * np.float32(_SQRT3)).astype(np.float32)


def synth_ssl(i: int, n_frames: int, seed: int = 1234) -> np.ndarray:
    """CN-HuBERT last_hidden_state stand-in, channels-first [1, 768, n_frames]."""
    return (hashed_uniform("ssl%d" % i, (1, 768, n_frames), seed) * np.float32(_SQRT3)).astype(np.float32)
