"""TEST INFRASTRUCTURE -- Python face of the CPU oracle (see gsv_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The heavy arithmetic is the C restatement in gsv_oracle.c (loaded through ctypes); the
control flow below restates the reference's AR drivers and sampling in numpy:

  sampling        gsv_tts/GPT_SoVITS/GPT/utils.py:5-59
  embeddings/PE   gsv_tts/GPT_SoVITS/GPT/embedding.py:52-75, t2s_model.py:351-361
  masks           t2s_model.py:300-349 (batched), 365-381 (single)
  infer           t2s_model.py:385-464
  infer_batched   t2s_model.py:555-734
  flow_dec        SoVITS/models.py:380-383

Parity status: pinned to the imported reference by oracle/gen_golden.py (fixtures in
tests/golden/); the reference itself ships no tests or golden vectors.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int)
c_u8 = ctypes.POINTER(ctypes.c_uint8)
c_i64 = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgsv_oracle.so")
    src = os.path.join(_HERE, "gsv_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgsv_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgsv_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_layer_floats.restype = ctypes.c_long
        _LIB.orc_flow_layer_floats.restype = ctypes.c_long
    return _LIB


def _fp(a):
    return a.ctypes.data_as(c_f)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int):
    lib().orc_set_num_threads(int(n))


# ----------------------------------------------------------------------------- sampling

def logits_to_probs(logits, previous_tokens=None, temperature=1.0, top_k=None, top_p=None,
                    repetition_penalty=1.0):
    """GPT/utils.py:12-49.  logits float32 [B, V] (modified in place like the reference)."""
    if previous_tokens is not None and repetition_penalty != 1.0:
        score = np.take_along_axis(logits, previous_tokens, axis=1)
        score = np.where(score < 0, score * np.float32(repetition_penalty),
                         score / np.float32(repetition_penalty)).astype(np.float32)
        np.put_along_axis(logits, previous_tokens, score, axis=1)
    if top_p is not None and top_p < 1.0:
        order = np.argsort(-logits, axis=1, kind="stable")
        sl = np.take_along_axis(logits, order, axis=1)
        e = np.exp(sl - sl.max(axis=1, keepdims=True))
        cum = np.cumsum(e / e.sum(axis=1, keepdims=True), axis=1)
        rem = cum > top_p
        rem[:, 0] = False
        remove = np.zeros_like(rem)
        np.put_along_axis(remove, order, rem, axis=1)
        logits = np.where(remove, -np.inf, logits).astype(np.float32)
    logits = (logits / np.float32(max(temperature, 1e-5))).astype(np.float32)
    if top_k is not None:
        k = min(top_k, logits.shape[-1])
        pivot = np.sort(logits, axis=1)[:, -k][:, None]
        logits = np.where(logits < pivot, -np.inf, logits).astype(np.float32)
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def sample(logits, previous_tokens=None, q=None, **kw):
    """GPT/utils.py:52-59.  q: exponential(1) noise [B, V]; None -> ones, which equals the
    reference whenever the kept set has a unique maximum (top_k=1)."""
    probs = logits_to_probs(logits, previous_tokens, **kw)
    if q is None:
        q = np.ones_like(probs)
    return np.argmax(probs / q, axis=-1)[:, None].astype(np.int64)


# ----------------------------------------------------------------------------- GPT

_LAYER_KEYS = ["qkv.weight", "qkv.bias", "out_proj.weight", "out_proj.bias", "norm1.weight",
               "norm1.bias", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias",
               "norm2.weight", "norm2.bias"]


def sine_pe(n_pos: int, dim: int) -> np.ndarray:
    """embedding.py:52-69, fp32."""
    pos = np.arange(0, n_pos, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, dim, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / dim)))
    pe = np.zeros((n_pos, dim), dtype=np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe


R_KV, R_LIN, R_ATTN, R_FP8, R_VOC, R_PART, R_FINE, R_FFN32 = 1, 2, 4, 8, 16, 32, 64, 128     # gsv_oracle.c ORC_R_*


def round_bf16(a):
    """fp32 -> nearest-even bf16 -> fp32 (what the product's bf16 weight / cache storage holds)"""
    a = np.ascontiguousarray(a, np.float32).copy()
    lib().orc_round_bf16(_fp(a), ctypes.c_long(a.size))
    return a


def round_e4m3(a):
    """fp32 -> OCP e4m3fn (nearest even, saturating at 448) -> fp32"""
    a = np.ascontiguousarray(a, np.float32).copy()
    lib().orc_round_e4m3(_fp(a), ctypes.c_long(a.size))
    return a


def quant_e4m3_rows(w):
    """the product's fp8 weight format: per output channel scale = max|row| / 448, row / scale -> e4m3; returns the
    dequantised weights (what the contraction multiplies, up to the epilogue's scale factor) and the scales"""
    w = np.ascontiguousarray(w, np.float32)
    amax = np.abs(w).max(axis=1, keepdims=True)
    scale = np.where(amax > 0, amax * np.float32(1.0 / 448.0), np.float32(1.0)).astype(np.float32)
    q = round_e4m3(w * (np.float32(1.0) / scale))
    return q, scale[:, 0]


class T2SOracle:
    """numerics: "fp32" = the reference's CPU arithmetic (the parity target);
    "bf16" / "fp8" = the same restatement with the product's production roundings applied at the same places
    (weights, K/V, GEMM operands -- module header of gsv_oracle.c), so that only the summation order separates
    it from the HIP kernels: the tight pin of the kernels the bench times.  `batched_min` mirrors the library's
    switch to the batched decode step (gsv_t2s_batched_min), `ffn_slices` its FFN slice count per batch size
    (gsv_t2s_ffn_slices: 64 at <= 4 sequences, else 32)."""

    FFN_SINGLE_MAX_B = 8      # the library's one-sequence-per-block FFN kernel up to here (gsv_abi.hip t2s_launch_ffn), two per block above

    def __init__(self, config, weights, gpt_cache, numerics="fp32", batched_min=12, ffn_slices=None):
        m = config["model"]
        self.D, self.H, self.NL = m["hidden_dim"], m["head"], m["n_layer"]
        self.V, self.EOS = m["vocab_size"], m["EOS"]
        self.suppressed = [280, 486, self.EOS]
        assert numerics in ("fp32", "bf16", "fp8")
        self.numerics = numerics
        self.batched_min = batched_min
        self.ffn_slices = ffn_slices or (lambda bsz: 64 if bsz <= 4 else 32)
        w = {k: _f32(v) for k, v in weights.items()}
        self.w8 = None
        if numerics != "fp32":
            w = dict(w)
            for k in list(w):
                if k.endswith((".qkv.weight", ".out_proj.weight", ".mlp.0.weight", ".mlp.2.weight")) or \
                        k in ("ar_predict_layer.weight", "bert_proj.weight"):
                    w[k] = round_bf16(w[k])
        self.w = w
        if numerics == "fp8":
            # second pack for the batched step: qkv / mlp weights dequantised from e4m3 with the row scale folded back
            w8 = dict(w)
            for k in list(w8):
                if k.endswith((".qkv.weight", ".mlp.0.weight", ".mlp.2.weight")):
                    q, sc = quant_e4m3_rows(_f32(weights[k]))
                    w8[k] = (q * sc[:, None]).astype(np.float32)
            self.pack8 = np.concatenate([w8["t2s_transformer.blocks.%d.%s" % (l, k)].ravel()
                                         for l in range(self.NL) for k in _LAYER_KEYS])
        self.pack = np.concatenate([w["t2s_transformer.blocks.%d.%s" % (l, k)].ravel()
                                    for l in range(self.NL) for k in _LAYER_KEYS])
        assert self.pack.size == lib().orc_layer_floats(self.D) * self.NL
        pe = sine_pe(4000, self.D)
        self.pe_text = (w["ar_text_position.alpha"][0] * pe).astype(np.float32)
        self.pe_audio = (w["ar_audio_position.alpha"][0] * pe).astype(np.float32)
        self.buckets = {}
        for b, t in gpt_cache:
            self.buckets.setdefault(b, []).append(t)
        for b in self.buckets:
            self.buckets[b].sort()
        self.cache = {}
        for b, ts in self.buckets.items():
            shape = (self.NL, b, self.H, ts[-1], self.D // self.H)
            self.cache[b] = (np.zeros(shape, np.float32), np.zeros(shape, np.float32))
        self.margins = []
        self.raw_margins = []

    # -- pieces -------------------------------------------------------------------------
    def embed_text(self, x, bert):
        w = self.w
        e = w["ar_text_embedding.word_embeddings.weight"][x]
        if self.numerics != "fp32":
            bert = round_bf16(bert)
        e = e + (bert @ w["bert_proj.weight"].T + w["bert_proj.bias"])
        return (e * np.float32(1.0) + self.pe_text[: len(x)]).astype(np.float32)

    def embed_audio(self, y):
        e = self.w["ar_audio_embedding.word_embeddings.weight"][y]
        return (e * np.float32(1.0) + self.pe_audio[: len(y)]).astype(np.float32)

    def next_input(self, tok, pos):
        """emb[tok]*1.0 + (alpha*pe)[pos]   t2s_model.py:420,456 (negative pos wraps like torch)."""
        e = self.w["ar_audio_embedding.word_embeddings.weight"][tok]
        return (e * np.float32(1.0) + self.pe_audio[pos]).astype(np.float32)

    @staticmethod
    def single_mask(lx, ly):
        L = lx + ly
        m = np.zeros((L, L), np.uint8)
        m[:lx, :lx] = 1
        m[lx:, :lx] = 1
        m[lx:, lx:] = np.tril(np.ones((ly, ly), np.uint8))
        return m

    def logits(self, h):
        # C path (not numpy/BLAS): a second thread pool inside the AR loop fights OpenMP's
        h = np.ascontiguousarray(h, np.float32)
        out = np.empty((h.shape[0], self.V), np.float32)
        lib().orc_linear(_fp(h), h.shape[0], self.D, _fp(self.w["ar_predict_layer.weight"]), None, self.V,
                         _fp(out), 0)
        return out

    def prefill(self, xy, mask, bsz, b0, nb=None):
        """xy [B, L, D], mask [B, L, L] -> hidden [B, L, D]; K/V into cache rows b0.."""
        kc, vc = self.cache[bsz]
        xy = np.ascontiguousarray(xy, np.float32).copy()
        mask = np.ascontiguousarray(mask, np.uint8)
        B, L, _ = xy.shape
        if L > kc.shape[3]:
            raise ValueError("prompt longer than the largest KV bucket")
        self._prefill_rows(xy, mask, bsz, b0)
        return xy

    def _prefill_rows(self, xy, mask, bsz, b0):
        """in place; the prompt path rounds K/V, GEMM operands and the attention's q / p in the reduced modes"""
        kc, vc = self.cache[bsz]
        lib().orc_set_rounding(0 if self.numerics == "fp32" else (R_KV | R_LIN | R_ATTN))
        try:
            lib().orc_t2s_prefill(_fp(self.pack), self.NL, self.D, self.H, xy.shape[0], xy.shape[1], _fp(xy),
                                  mask.ctypes.data_as(c_u8), _fp(kc), _fp(vc), kc.shape[1], kc.shape[3], int(b0))
        finally:
            lib().orc_set_rounding(0)

    def decode(self, x, bsz, kv_len):
        kc, vc = self.cache[bsz]
        x = np.ascontiguousarray(x, np.float32).copy()
        kv = np.ascontiguousarray(kv_len, np.int64)
        pack, flags = self.pack, 0
        if self.numerics != "fp32":
            flags = R_KV
            if bsz >= self.batched_min and kc.shape[3] <= 1024:   # batched chain (gsv_t2s_decode takes it up to 1024 cache positions): bf16 (fp8) MFMA operands
                flags |= R_LIN
                if self.numerics == "fp8":
                    flags |= R_FP8
                    pack = self.pack8
            else:                                          # partial-sum kernels: head / slice partials cross the boundary as fp16
                flags |= R_PART
                n_sl = int(self.ffn_slices(bsz))
                assert n_sl in (32, 64)
                if n_sl == 64:
                    flags |= R_FINE
                # which dots take ONE bf16 value per activation (round 5) and which still run fp32 FMA chains on unpacked weights:
                #   <= 8 sequences      t2s_attn_kernel + t2s_ffn_kernel: bf16 activations everywhere
                #   9 .. 16             t2s_attn_kernel (bf16 activations) + t2s_ffn_multi_kernel<.., 2> (fp32 activations)
                #   > 16 (long caches)  t2s_attn_multi_kernel + t2s_ffn_multi_kernel: fp32 activations everywhere
                if bsz <= self.FFN_SINGLE_MAX_B:
                    flags |= R_LIN
                elif bsz <= 16:
                    flags |= R_LIN | R_FFN32
        lib().orc_set_rounding(flags)
        try:
            lib().orc_t2s_decode(_fp(pack), self.NL, self.D, self.H, x.shape[0], _fp(x), _fp(kc), _fp(vc),
                                 kc.shape[1], kc.shape[3], 0, kv.ctypes.data_as(c_i64))
        finally:
            lib().orc_set_rounding(0)
        return x

    @staticmethod
    def _gap(logits_row):
        t = np.sort(logits_row[np.isfinite(logits_row)])
        return float(t[-1] - t[-2])

    def _margin(self, logits_row):
        """decision margin: top-1 minus top-2 of the logits the argmax actually saw."""
        t = np.sort(logits_row[np.isfinite(logits_row)])
        self.margins.append(float(t[-1] - t[-2]))

    def _raw(self, lg):
        t = np.sort(lg, axis=-1)
        self.raw_margins.append(float((t[:, -1] - t[:, -2]).min()))

    # -- drivers ------------------------------------------------------------------------
    def infer(self, x, y, bert, top_k=15, top_p=1.0, temperature=1.0, repetition_penalty=1.35,
              initial_suppression_steps=10, check_interval=5, rng=None):
        x = np.asarray(x, np.int64); y = np.asarray(y, np.int64)
        lx, ly = len(x), len(y)
        L = lx + ly
        xy = np.concatenate([self.embed_text(x, _f32(bert)), self.embed_audio(y)])[None]
        bks = self.buckets[1]
        bi = len(bks) - 1
        for i, t in enumerate(bks):
            if t > L:
                bi = i
                break
        kw = dict(top_k=top_k, top_p=top_p, temperature=temperature, repetition_penalty=repetition_penalty)
        q = (lambda shape: rng.exponential(size=shape).astype(np.float32)) if rng is not None else (lambda s: None)
        self.margins = []
        self.raw_margins = []
        h = self.prefill(xy, self.single_mask(lx, ly)[None], 1, 0)
        kv = L
        lg = self.logits(h[:, -1])
        self._raw(lg)
        lg[:, self.suppressed] = -np.inf
        pre = y[None].copy()
        view = lg[:, :-1].copy()
        s = sample(view, pre, q=q(view.shape), **kw)
        self._margin(view[0])
        pre = np.concatenate([pre, s], axis=1)
        xin = self.next_input(s[:, 0], kv - lx)
        n_iter = bks[-1] - kv
        idx = 0
        for idx in range(1, n_iter + 1):
            if kv == bks[bi]:
                bi += 1
            h = self.decode(xin, 1, [kv])
            kv += 1
            lg = self.logits(h)
            self._raw(lg)
            if idx < initial_suppression_steps:
                lg[:, self.suppressed] = -np.inf
            s = sample(lg, pre, q=q(lg.shape), **kw)
            self._margin(lg[0])
            pre = np.concatenate([pre, s], axis=1)
            if idx % check_interval == 0 and s[0, 0] == self.EOS:
                break
            xin = self.next_input(s[:, 0], kv - lx)
        if idx == 0:
            raise RuntimeError("no decode iterations: prompt fills the largest bucket")
        out = pre[0, -idx:]
        e = np.nonzero(out == self.EOS)[0]
        return out[: e[0]] if e.size else out

    def infer_stream(self, x, y, bert, top_k=15, top_p=1.0, temperature=1.0, repetition_penalty=1.35,
                     initial_suppression_steps=10, stream_chunk=25, boost_first_chunk=True, rng=None):
        """t2s_model.py:466-553 as a generator of (cumulative tokens, is_final).  Differences from infer() that
        are the reference's: EOS is tested on EVERY step and is never appended; chunks are cumulative and lag one
        chunk behind (the first one is sent at once when boost_first_chunk); the final yield is the last `idx`
        entries of y ++ samples, which after an EOS break still contains the first sample s0."""
        x = np.asarray(x, np.int64); y = np.asarray(y, np.int64)
        lx, ly = len(x), len(y)
        L = lx + ly
        xy = np.concatenate([self.embed_text(x, _f32(bert)), self.embed_audio(y)])[None]
        bks = self.buckets[1]
        kw = dict(top_k=top_k, top_p=top_p, temperature=temperature, repetition_penalty=repetition_penalty)
        q = (lambda shape: rng.exponential(size=shape).astype(np.float32)) if rng is not None else (lambda s: None)
        h = self.prefill(xy, self.single_mask(lx, ly)[None], 1, 0)
        kv = L
        lg = self.logits(h[:, -1])
        lg[:, self.suppressed] = -np.inf
        pre = y[None].copy()
        view = lg[:, :-1].copy()
        s = sample(view, pre, q=q(view.shape), **kw)
        pre = np.concatenate([pre, s], axis=1)
        xin = self.next_input(s[:, 0], kv - lx)
        first, pre_chunk, idx = True, None, 0
        for idx in range(1, bks[-1] - kv + 1):
            h = self.decode(xin, 1, [kv])
            kv += 1
            lg = self.logits(h)
            if idx < initial_suppression_steps:
                lg[:, self.suppressed] = -np.inf
            s = sample(lg, pre, q=q(lg.shape), **kw)
            if s[0, 0] == self.EOS:
                break
            pre = np.concatenate([pre, s], axis=1)
            if idx % stream_chunk == 0:
                if pre_chunk is not None:
                    yield pre_chunk, False
                pre_chunk = pre[0, -idx:].copy()
                if boost_first_chunk and first:
                    first = False
                    yield pre_chunk, False
                    pre_chunk = None
            xin = self.next_input(s[:, 0], kv - lx)
        yield pre[0, -idx:].copy(), True

    def infer_batched(self, xs, ys, berts, top_k=15, top_p=1.0, temperature=1.0,
                      repetition_penalty=1.35, check_interval=5, rng=None):
        B = len(xs)
        sizes = sorted(self.buckets)
        bsz = sizes[-1]
        for s_ in sizes:
            if s_ >= B:
                bsz = s_
                break
        actual = min(B, bsz)
        kw = dict(top_k=top_k, top_p=top_p, temperature=temperature, repetition_penalty=repetition_penalty)
        q = (lambda shape: rng.exponential(size=shape).astype(np.float32)) if rng is not None else (lambda s: None)
        x_lens = np.array([len(a) for a in xs[:bsz]], np.int64)
        y_lens = np.array([len(a) for a in ys[:bsz]], np.int64)
        xy_lens = x_lens + y_lens
        Lmax = int(xy_lens.max())
        xy = np.zeros((actual, Lmax, self.D), np.float32)
        mask = np.zeros((actual, Lmax, Lmax), np.uint8)
        for b in range(actual):
            lx, ly = int(x_lens[b]), int(y_lens[b])
            xy[b, :lx] = self.embed_text(np.asarray(xs[b], np.int64), _f32(berts[b]))
            xy[b, lx:lx + ly] = self.embed_audio(np.asarray(ys[b], np.int64))
            mask[b, :lx + ly, :lx + ly] = self.single_mask(lx, ly)
        bks = self.buckets[bsz]
        bi = len(bks) - 1
        for i, t in enumerate(bks):
            if t > Lmax:
                bi = i
                break
        kv = np.zeros(bsz, np.int64)
        cur = actual
        pre = np.zeros((bsz, bks[-1]), np.int64)
        h = self.prefill(xy, mask, bsz, 0)
        last = h[np.arange(actual), xy_lens[:actual] - 1]
        lg = self.logits(last)
        kv[:actual] = xy_lens
        view = lg[:, :-1].copy()
        samples = sample(view, None, q=q(view.shape), **kw)
        # decision margins per request (top-1 minus top-2 of the logits each of its samples saw); entry 0 = prefill sample
        self.req_margins = {b: [self._gap(view[b])] for b in range(actual)}
        xin = np.zeros((bsz, self.D), np.float32)
        xin[:actual] = self.next_input(samples[:, 0], kv[:actual] - x_lens)
        x_lens = np.concatenate([x_lens, np.zeros(bsz - actual, np.int64)])
        samples = np.concatenate([samples, np.zeros((bsz - actual, 1), np.int64)])
        pred, orig = [], []
        slot_orig = np.arange(bsz)
        steps = np.zeros(bsz, np.int64)
        ignore = np.ones(bsz, bool)
        ignore[:actual] = False
        stop = False
        rows = np.arange(bsz)
        while True:
            for idx in range(1000):
                steps += 1
                h = self.decode(xin, bsz, kv)
                kv += 1
                lg = self.logits(h)
                samples = sample(lg, None, q=q(lg.shape), **kw)
                for i in np.nonzero(~ignore)[0]:
                    self.req_margins[int(slot_orig[i])].append(self._gap(lg[i]))
                pre[rows, kv] = samples[:, 0]
                if idx % check_interval == 0:
                    reached = kv + check_interval >= bks[min(bi, len(bks) - 1)]
                    eos = samples[:, 0] == self.EOS
                    fin = ~ignore & (eos | reached)
                    if fin.any():
                        if reached.any():
                            bi += 1
                            if bi < len(bks):
                                reached[:] = False
                        fin = ~ignore & (eos | reached)
                        if fin.any():
                            for i in np.nonzero(fin)[0]:
                                seg = pre[i, kv[i] - steps[i] + 1: kv[i]]
                                e = np.nonzero(seg == self.EOS)[0]
                                if e.size:
                                    seg = seg[: e[0]]
                                pred.append(seg.copy())
                                orig.append(int(slot_orig[i]))
                                steps[i] = 0
                                kv[i] = 0
                                mx = int(kv.max())
                                bi = len(bks) - 1
                                for j, t in enumerate(bks):
                                    if t >= mx + check_interval:
                                        bi = j
                                        break
                                if cur == B:
                                    ignore[i] = True
                                    if ignore.all():
                                        stop = True
                                        break
                                else:
                                    sx = np.asarray(xs[cur], np.int64); sy = np.asarray(ys[cur], np.int64)
                                    one = np.concatenate([self.embed_text(sx, _f32(berts[cur])),
                                                          self.embed_audio(sy)])[None]
                                    xyd = np.ascontiguousarray(one, np.float32)
                                    m1 = self.single_mask(len(sx), len(sy))[None]
                                    self._prefill_rows(xyd, m1, bsz, int(i))
                                    l1 = self.logits(xyd[:, -1])
                                    x_lens[i] = len(sx)
                                    kv[i] = len(sx) + len(sy)
                                    v1 = l1[:, :-1].copy()
                                    samples[i: i + 1] = sample(v1, None, q=q(v1.shape), **kw)
                                    self.req_margins[cur] = [self._gap(v1[0])]
                                    slot_orig[i] = cur
                                    cur += 1
                            if stop:
                                break
                xin = self.next_input(samples[:, 0], kv - x_lens)
            if stop:
                break
        return pred, np.array(orig, np.int64)


# ----------------------------------------------------------------------------- SoVITS flow + Generator

def fold_weight_norm(g, v):
    """torch.nn.utils.weight_norm: W = v * (g / ||v||), norm over all dims but 0."""
    g = _f32(g); v = _f32(v)
    n = np.sqrt((v * v).sum(axis=tuple(range(1, v.ndim)), keepdims=True, dtype=np.float32))
    return (v * (g / n)).astype(np.float32)


class VocoderOracle:
    """numerics "bf16": conv weights (flow: after the weight-norm fold) rounded to bf16 and every stored activation
    rounded where the bf16 HIP path stores bf16 (ORC_R_VOC); conv_post keeps fp32 weights, as the product does."""

    def __init__(self, hps, weights, numerics="fp32"):
        assert numerics in ("fp32", "bf16")
        self.numerics = numerics
        m = hps["model"]
        self.H = m["hidden_channels"]; self.inter = m["inter_channels"]; self.gin = m["gin_channels"]
        self.C0 = m["upsample_initial_channel"]
        self.up_rates = np.array(m["upsample_rates"], np.int32)
        self.up_kernels = np.array(m["upsample_kernel_sizes"], np.int32)
        self.rk = np.array(m["resblock_kernel_sizes"], np.int32)
        self.rdil = np.array(m["resblock_dilation_sizes"][0], np.int32)
        self.samples_per_frame = int(np.prod(self.up_rates))
        w = {k: _f32(v) for k, v in weights.items()}
        parts = [w["dec.conv_pre.weight"], w["dec.conv_pre.bias"], w["dec.cond.weight"], w["dec.cond.bias"]]
        for i in range(len(self.up_rates)):
            parts += [w["dec.ups.%d.weight" % i], w["dec.ups.%d.bias" % i]]
            for j in range(len(self.rk)):
                r = "dec.resblocks.%d." % (i * len(self.rk) + j)
                for c in ("convs1", "convs2"):
                    for d in range(3):
                        parts += [w["%s%s.%d.weight" % (r, c, d)], w["%s%s.%d.bias" % (r, c, d)]]
        if numerics == "bf16":
            parts = [p if p.ndim == 1 else round_bf16(p) for p in parts]      # weights, not biases
        parts.append(w["dec.conv_post.weight"])
        self.gen_pack = np.concatenate([p.ravel() for p in parts])
        fparts = []
        for fl in range(0, 8, 2):
            p = "flow.flows.%d." % fl
            fparts += [w[p + "pre.weight"], w[p + "pre.bias"],
                       fold_weight_norm(w[p + "enc.cond_layer.weight_g"], w[p + "enc.cond_layer.weight_v"]),
                       w[p + "enc.cond_layer.bias"]]
            for l in range(4):
                fparts += [fold_weight_norm(w["%senc.in_layers.%d.weight_g" % (p, l)], w["%senc.in_layers.%d.weight_v" % (p, l)]),
                           w["%senc.in_layers.%d.bias" % (p, l)],
                           fold_weight_norm(w["%senc.res_skip_layers.%d.weight_g" % (p, l)], w["%senc.res_skip_layers.%d.weight_v" % (p, l)]),
                           w["%senc.res_skip_layers.%d.bias" % (p, l)]]
            fparts += [w[p + "post.weight"], w[p + "post.bias"]]
        if numerics == "bf16":
            fparts = [p if p.ndim == 1 else round_bf16(p) for p in fparts]
        self.flow_pack = np.concatenate([p.ravel() for p in fparts])
        assert self.flow_pack.size == 4 * lib().orc_flow_layer_floats(self.inter // 2, self.H, self.gin)

    def flow(self, z_p, y_mask, ge):
        """z_p [C, T], y_mask [T], ge [gin, Tg] -> [C, T]  (ResidualCouplingBlock reverse)."""
        x = _f32(z_p).copy()
        m = _f32(y_mask).ravel()
        g = _f32(ge).reshape(self.gin, -1)
        lib().orc_set_rounding(R_VOC if self.numerics == "bf16" else 0)
        try:
            lib().orc_flow_reverse(_fp(self.flow_pack), 4, self.inter // 2, self.H, self.gin, _fp(x), x.shape[1],
                                   _fp(m), _fp(g), g.shape[1])
        finally:
            lib().orc_set_rounding(0)
        return x

    def dec(self, z, ge):
        z = _f32(z)
        g = _f32(ge).reshape(self.gin, -1)
        T = z.shape[1]
        out = np.zeros(T * self.samples_per_frame, np.float32)
        lib().orc_set_rounding(R_VOC if self.numerics == "bf16" else 0)
        try:
            lib().orc_generator(_fp(self.gen_pack), self.inter, self.C0, self.gin, len(self.up_rates),
                                self.up_rates.ctypes.data_as(c_i), self.up_kernels.ctypes.data_as(c_i),
                                len(self.rk), self.rk.ctypes.data_as(c_i), self.rdil.ctypes.data_as(c_i),
                                _fp(z), T, _fp(g), g.shape[1], _fp(out))
        finally:
            lib().orc_set_rounding(0)
        return out

    def flow_dec(self, z_p, y_mask, ge):
        """models.py:380-383: o = dec(flow(z_p, mask, ge) * mask, g=ge)."""
        z = self.flow(z_p, y_mask, ge)
        return self.dec(z * _f32(y_mask).reshape(1, -1), ge)


# ---- restatement of the DEVICE sampler (csrc/t2s_decode.h: t2s_uniform + the ctl[0] == 2 branch of the
# token kernel).  The reference draws Exp(1) noise from torch's generator (GPT/utils.py:56-59); the device
# sampler owns a counter-based stream instead, so the checker restates that stream, not torch's.
def _lowbias32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16); h *= np.uint32(0x7feb352d); h ^= h >> np.uint32(15); h *= np.uint32(0x846ca68b); h ^= h >> np.uint32(16)
    return h


def device_uniform(seed, slot, pos, step, V):
    lo, hi = np.uint32(seed & 0x7fffffff), np.uint32((seed >> 31) & 0x7fffffff)
    v = np.arange(V, dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = lo ^ (v * np.uint32(0x9E3779B1)) ^ (np.uint32(pos) * np.uint32(0x85EBCA77)) ^ (np.uint32(slot) * np.uint32(0xC2B2AE3D)) \
            ^ (np.uint32(step) * np.uint32(0x27D4EB2F)) ^ ((hi << np.uint32(13)) | (hi >> np.uint32(19)))
        h = _lowbias32(h)
        h = h + hi
        h = _lowbias32(h)
    return ((h >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def device_normal(seed, n):
    """the noise of gsv_voc_decode (csrc/gsv_voc.hip dec_normal): element i of the stream `seed` (uint64) -- Box-Muller over two
    counter-based uniforms; returned in float64 (the device evaluates log / cos / sqrt in float32: compare with a tolerance)"""
    lo, hi = np.uint32(seed & 0xffffffff), np.uint32((seed >> 32) & 0xffffffff)
    i = np.arange(n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        a = _lowbias32(_lowbias32(i * np.uint32(0x9E3779B1) ^ lo) + hi)
        b = _lowbias32(_lowbias32(i * np.uint32(0x85EBCA77) ^ hi ^ np.uint32(0x68E31DA4)) + lo)
    u1 = ((a >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
    u2 = ((b >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _device_top_p(logits, top_p):
    """the device sampler's top-p: keep v iff sum_{p_u >= p_v} p_u <= top_p, or v is the arg-max (== the reference's
    sort + cumsum rule, GPT/utils.py:29-40, with ties kept or dropped together)"""
    x = np.asarray(logits, np.float32).copy()
    if top_p is None or not (0.0 < top_p < 1.0):
        return x
    fin = np.isfinite(x)
    p = np.where(fin, np.exp(x.astype(np.float64) - x[fin].max()), 0.0)
    p /= p.sum()
    order = np.argsort(-p, kind="stable")
    cum = np.cumsum(p[order])
    g = np.empty_like(p)
    ps = p[order]
    # inclusive mass with ties grouped: the cumulative value at the LAST position of each run of equal probabilities
    last = np.r_[np.nonzero(np.diff(ps))[0], len(ps) - 1]
    g[order] = np.repeat(cum[last], np.diff(np.r_[-1, last]))
    keep = (g <= top_p + 1e-12) | (np.arange(len(x)) == int(np.argmax(x)))
    x[~keep] = -np.inf
    return x


def device_sample(logits, top_k, temperature, seed, slot, pos, step, top_p=1.0):
    """(token, margin): top-p, temperature, top-k with ties kept, Gumbel argmax; margin = top-1 minus top-2 score"""
    x = _device_top_p(logits, top_p) / np.float32(max(temperature, 1e-5))
    V = x.shape[0]
    pivot = -np.inf
    if top_k and 0 < top_k < V:
        pivot = np.sort(x)[::-1][top_k - 1]
    keep = (x >= pivot) & np.isfinite(x)
    u = device_uniform(seed, slot, pos, step, V)
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(keep, (x - x.max()) - np.log(-np.log(u)), -np.inf).astype(np.float32)
    order = np.argsort(-sc, kind="stable")
    return int(order[0]), float(sc[order[0]] - sc[order[1]]) if V > 1 else np.inf


def device_sample_probs(logits, top_k, temperature, top_p=1.0):
    """the distribution the device sampler draws from (== GPT/utils.py logits_to_probs)"""
    x = _device_top_p(logits, top_p).astype(np.float64) / max(temperature, 1e-5)
    if top_k and 0 < top_k < x.shape[0]:
        pivot = np.sort(x)[::-1][top_k - 1]
        x = np.where(x < pivot, -np.inf, x)
    e = np.exp(x - x[np.isfinite(x)].max())
    return e / e.sum()


# ---------------------------------------------------------------------------------------------
# subtitle alignment (TTS._viterbi_monotonic, gsv_tts/TTS.py:1744-1797)
# ---------------------------------------------------------------------------------------------
def viterbi_normal(attn):
    """head-averaged attention used by the alignment, TTS.py:1748-1767: heads whose arg-max is the last phoneme
    do not vote; frames without votes get the fixed near-uniform row.  That row's renormalising sum is taken
    in fp64 (closed form): torch's fp32 row sum depends on the host's vector width, so the reference has no
    device-independent value for that one scalar."""
    a = _f32(attn)
    H, T, N = a.shape
    votes = a.argmax(-1) != N - 1                                    # [H, T]
    count = votes.sum(0)
    s = np.zeros((T, N), np.float32)
    for h in range(H):
        s = s + a[h] * votes[h][:, None].astype(np.float32)
    f1, f09, f11 = np.float32(1.0 / N), np.float32(0.9 / N), np.float32(1.1 / N)
    row = np.full(N, f1, np.float32)
    row[N - 1] = f09
    row[1] = f11
    dsum = (np.float64(N - 2) * np.float64(f1) + np.float64(f09) + np.float64(f11)) if N > 2 else np.float64(f1) + np.float64(f11)
    row = row / np.float32(dsum)
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = s / count.astype(np.float32)[:, None]
    return np.where(count[:, None] > 0, mean, row[None, :]).astype(np.float32)


def viterbi_monotonic(attn):
    """attn [H, T, N] -> int64 [T]; TTS.py:1744-1797 (dp over frames, stay-or-advance-by-one, ties stay)."""
    normal = viterbi_normal(attn)
    T, N = normal.shape
    peak0 = normal.argmax(-1) == 0
    first_zero = int(np.nonzero(peak0)[0][0]) if peak0.any() else 0
    dp = normal[0].copy()
    adv = np.zeros((T, N), bool)
    for t in range(1, T):
        shifted = np.concatenate([np.float32([-np.inf]), dp[:-1]])
        adv[t] = shifted > dp
        dp = normal[t] + np.where(adv[t], shifted, dp)
    path = np.zeros(T, np.int64)
    path[-1] = int(dp.argmax())
    for t in range(T - 2, -1, -1):
        path[t] = path[t + 1] - int(adv[t + 1, path[t + 1]])
    path[:first_zero] = -1
    return path


# ---------------------------------------------------------------------------------------------
# reference-audio path
# ---------------------------------------------------------------------------------------------
def spectrogram(audio, n_fft=2048, hop=640):
    """torchaudio.transforms.Spectrogram(n_fft, win_length=n_fft, hop_length=hop, center=True, pad_mode="reflect",
    power=1.0) as TTS._get_spec builds it (gsv_tts/TTS.py:1591-1604).  torchaudio is a third-party dependency that
    is absent from /root/reference and from this image (the reference pins no version); its published algorithm is
    |torch.stft(x, n_fft, hop, n_fft, window=hann_window(n_fft) [periodic], center=True, pad_mode="reflect",
    normalized=False, onesided=True)|, restated here with an fp64 FFT.  -> float32 [n_fft/2+1][1 + n//hop]"""
    x = np.asarray(audio, np.float64).reshape(-1)
    xp = np.pad(x, n_fft // 2, mode="reflect")
    T = 1 + x.shape[0] // hop
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    frames = np.stack([xp[t * hop:t * hop + n_fft] for t in range(T)]) * w
    return np.abs(np.fft.rfft(frames, axis=1)).T.astype(np.float32)


class RefAudioOracle:
    """SynthesizerTrn.get_ge (SoVITS/models.py:371-378; MelStyleEncoder, module/modules.py:367-444) and
    extract_latent (models.py:431-434; EuclideanCodebook.quantize, module/core_vq.py:124-128) in numpy fp32."""

    def __init__(self, weights):
        self.w = {k: _f32(v) for k, v in weights.items()}

    def _lin(self, x, name):
        return x @ self.w[name + ".weight"].T + self.w[name + ".bias"]

    @staticmethod
    def _mish(x):
        sp = np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(np.float32)
        return x * np.tanh(sp)

    def get_ge(self, spec, sv_emb=None):
        """spec [bins >= 704][T] channels-first -> ge [gin]"""
        x = _f32(spec)[:704].T                                          # [T, 704]
        x = self._mish(self._lin(x, "ref_enc.spectral.0.fc"))
        x = self._mish(self._lin(x, "ref_enc.spectral.3.fc"))
        T = x.shape[0]
        for i in (0, 1):                                                # Conv1dGLU, modules.py:238-254
            W = self.w["ref_enc.temporal.%d.conv1.conv.weight" % i]     # [256, 128, 5]
            xp = np.pad(x, ((2, 2), (0, 0)))
            c = sum(xp[k:k + T] @ W[:, :, k].T for k in range(5)) + self.w["ref_enc.temporal.%d.conv1.conv.bias" % i]
            x = x + c[:, :128] * (1.0 / (1.0 + np.exp(-c[:, 128:])))
        q, k, v = (self._lin(x, "ref_enc.slf_attn." + n) for n in ("w_qs", "w_ks", "w_vs"))
        heads = []
        for h in range(2):                                              # modules.py:291-363
            s = q[:, h * 64:(h + 1) * 64] @ k[:, h * 64:(h + 1) * 64].T / np.float32(np.sqrt(128.0))
            s = np.exp(s - s.max(-1, keepdims=True))
            heads.append((s / s.sum(-1, keepdims=True)) @ v[:, h * 64:(h + 1) * 64])
        x = self._lin(np.concatenate(heads, 1), "ref_enc.slf_attn.fc") + x
        f = self._lin(x, "ref_enc.fc.fc")
        ge = (f / np.float32(T)).sum(0)                                 # temporal_avg_pool, modules.py:409-419
        if sv_emb is not None:                                          # models.py:374-377
            ge = ge + self._lin(_f32(sv_emb).reshape(1, -1), "sv_emb")[0]
            ge = np.where(ge >= 0, ge, self.w["prelu.weight"] * ge)
        return ge.astype(np.float32)

    def extract_latent(self, ssl):
        """ssl [768][Th] channels-first -> (codes int64 [Th//2], margin float32 [Th//2])"""
        x = _f32(ssl).T                                                 # [Th, 768]
        To = x.shape[0] // 2
        W = self.w["ssl_proj.weight"]                                   # [768, 768, 2]
        y = x[0:2 * To:2] @ W[:, :, 0].T + x[1:2 * To:2] @ W[:, :, 1].T + self.w["ssl_proj.bias"]
        e = self.w["quantizer.vq.layers.0._codebook.embed"]
        dist = -(((y * y).sum(1, keepdims=True) - 2 * (y @ e.T)) + (e * e).sum(1)[None, :])
        top = np.sort(dist, axis=1)
        return dist.argmax(1).astype(np.int64), (top[:, -1] - top[:, -2]).astype(np.float32)


# ------------------------------------------------------------------------------------------------ streaming splice
def sola(f1_overlap, f2, overlap_len, search_len=320):
    """TTS._sola_algorithm (gsv_tts/TTS.py:1612-1627) restated in numpy: slide the new chunk `f2` over the previous chunk's tail
    `f1_overlap` by the offset that maximises corr / sqrt(energy + 1e-8) (first maximum), cross-fade `overlap_len` samples with
    alpha = torch.linspace(0, 1, overlap_len) (torch's two-sided evaluation: start + step * j below the middle, end - step *
    (n - 1 - j) above).  -> (spliced chunk float32, offset).  Pinned to the reference by tests/golden/sola.npz."""
    f1 = np.asarray(f1_overlap, np.float32)
    f2 = np.asarray(f2, np.float32)
    key = f2[:overlap_len + search_len]
    n_off = len(key) - overlap_len + 1
    win = np.lib.stride_tricks.sliding_window_view(key, overlap_len)[:n_off]            # [n_off][overlap]
    corr = (win.astype(np.float64) * f1.astype(np.float64)).sum(1)
    energy = (win.astype(np.float64) ** 2).sum(1) + 1e-8
    score = (corr / np.sqrt(energy)).astype(np.float32)
    off = int(np.argmax(score))
    al = f2[off:]
    n = overlap_len
    step = np.float32(1.0) / np.float32(max(n - 1, 1))
    j = np.arange(n)
    alpha = np.where(j < n // 2, step * j.astype(np.float32), np.float32(1.0) - step * (n - 1 - j).astype(np.float32)).astype(np.float32)
    if n == 1:
        alpha = np.zeros(1, np.float32)
    faded = f1 * (np.float32(1.0) - alpha) + al[:n] * alpha
    return np.concatenate([faded, al[n:]]).astype(np.float32), off
