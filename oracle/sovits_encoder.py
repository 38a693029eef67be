"""TEST INFRASTRUCTURE (a checker, like the rest of oracle/): SoVITS text/ssl encoder `enc_p` (TextEncoder), codebook lookup
and the `decode()` wrapper around them, restated in plain torch.  Only tests/ import it; the product's `decode()` is one
C-ABI call (gsv_voc_decode) and never touches this file.  The restatement is pinned to the reference by
tests/golden/decode.npz (tests/test_hip_tts.py::test_decode_end_to_end_matches_reference runs the product against the
reference's own outputs; tests/test_hip_encp.py holds the device enc_p to this restatement).  Semantics follow the reference:

  TextEncoder.infer            gsv_tts/GPT_SoVITS/SoVITS/models.py:196-224
  attentions.Encoder / FFN     SoVITS/module/attentions.py:10-77, 221-277
  windowed relative attention  SoVITS/module/attentions.py:80-219  (window 4, shared heads)
  MRTE cross attention         SoVITS/module/mrte_model.py:6-38
  channel LayerNorm            SoVITS/module/modules.py:15-27
  codebook decode              SoVITS/module/core_vq.py:147-149,222-226,295-301

The relative-position terms are computed directly on the 2w+1 band (gather / scatter on the
offset j - i) instead of the reference's pad-and-reshape skewing; the result is the same
attention (embeddings outside the window are zero in both formulations).

`round_fn` (tests only; default: identity = the fp32 arithmetic of the reference): applied to the conv / linear
weights and to every tensor the bf16 device path (csrc/encp.h, gsv_voc.hip encp_run) stores as bf16 -- gathered
embeddings, conv outputs, attention probabilities and outputs, LayerNorm outputs, the FFN hidden -- so that
tests/test_hip_encp.py can hold the device enc_p to a rounding-matched mirror instead of the fp32 one.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _dev(weights, name, device):
    return weights[name].detach().to(device=device, dtype=torch.float32)


def _ident(x):
    return x


def codebook_decode(weights, codes):
    """codes int64 [n_q=1, B, N] -> [B, 768, N]"""
    out = None
    for q in range(codes.shape[0]):
        emb = weights["quantizer.vq.layers.%d._codebook.embed" % q]
        emb = emb.detach().to(device=codes.device, dtype=torch.float32)
        v = F.embedding(codes[q], emb).transpose(1, 2)
        out = v if out is None else out + v
    return out


def channel_layer_norm(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, eps).transpose(1, -1)


class _Attention:
    """multi-head attention over channels-first tensors; optional windowed relative positions"""

    def __init__(self, w, prefix, device, n_heads, window=None, rnd=_ident):
        g = lambda n: _dev(w, prefix + n, device)
        self.rnd = rnd
        self.wq, self.bq = rnd(g("conv_q.weight")), g("conv_q.bias")
        self.wk, self.bk = rnd(g("conv_k.weight")), g("conv_k.bias")
        self.wv, self.bv = rnd(g("conv_v.weight")), g("conv_v.bias")
        self.wo, self.bo = rnd(g("conv_o.weight")), g("conv_o.bias")
        self.n_heads = n_heads
        self.window = window
        if window is not None:
            self.rel_k, self.rel_v = g("emb_rel_k"), g("emb_rel_v")  # [1, 2w+1, dk]
        self.attn = None

    def __call__(self, x, c, mask):
        rnd = self.rnd
        q = rnd(F.conv1d(x, self.wq, self.bq))
        k = rnd(F.conv1d(c, self.wk, self.bk))
        v = rnd(F.conv1d(c, self.wv, self.bv))
        b, d, tt = q.shape
        ts = k.shape[2]
        h, dk = self.n_heads, d // self.n_heads
        q = q.view(b, h, dk, tt).transpose(2, 3) / math.sqrt(dk)
        k = k.view(b, h, dk, ts).transpose(2, 3)
        v = v.view(b, h, dk, ts).transpose(2, 3)
        scores = q @ k.transpose(-2, -1)
        if self.window is not None:
            w = self.window
            idx = torch.arange(ts, device=x.device)
            off = idx[None, :] - idx[:, None]                       # j - i
            band = off.abs() <= w
            bin_ = (off.clamp(-w, w) + w)[None, None].expand(b, h, tt, ts)
            rel_logits = q @ self.rel_k.transpose(-2, -1)           # [b,h,t,2w+1]
            scores = scores + torch.gather(rel_logits, -1, bin_) * band
        if mask is not None:
            scores = scores.masked_fill(mask == 0, -1e4)
        p = torch.softmax(scores, dim=-1)
        out = rnd(p) @ v
        if self.window is not None:
            rel_w = torch.zeros(b, h, tt, 2 * w + 1, device=x.device, dtype=p.dtype)
            rel_w.scatter_add_(-1, bin_, p * band)
            out = out + rel_w @ self.rel_v
        self.attn = p
        out = rnd(out.transpose(2, 3).contiguous().view(b, d, tt))
        return F.conv1d(out, self.wo, self.bo)


class _Encoder:
    def __init__(self, w, prefix, device, n_heads, n_layers, kernel_size, window=4, rnd=_ident):
        self.layers = []
        self.rnd = rnd
        for i in range(n_layers):
            g = lambda n: _dev(w, prefix + n, device)
            self.layers.append(dict(
                attn=_Attention(w, "%sattn_layers.%d." % (prefix, i), device, n_heads, window, rnd),
                n1=(g("norm_layers_1.%d.gamma" % i), g("norm_layers_1.%d.beta" % i)),
                n2=(g("norm_layers_2.%d.gamma" % i), g("norm_layers_2.%d.beta" % i)),
                c1=(rnd(g("ffn_layers.%d.conv_1.weight" % i)), g("ffn_layers.%d.conv_1.bias" % i)),
                c2=(rnd(g("ffn_layers.%d.conv_2.weight" % i)), g("ffn_layers.%d.conv_2.bias" % i)),
            ))
        self.k = kernel_size

    def __call__(self, x, x_mask):
        attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
        x = x * x_mask
        pl, pr = (self.k - 1) // 2, self.k // 2
        rnd = self.rnd
        for L in self.layers:
            y = L["attn"](x, x, attn_mask)
            x = rnd(channel_layer_norm(x + y, *L["n1"]))
            y = F.conv1d(F.pad(x * x_mask, (pl, pr)), *L["c1"])
            y = rnd(torch.relu(y))
            y = F.conv1d(F.pad(y * x_mask, (pl, pr)), *L["c2"]) * x_mask
            x = rnd(channel_layer_norm(x + y, *L["n2"]))
        return x * x_mask


class _MRTE:
    def __init__(self, w, device, rnd=_ident):
        g = lambda n: _dev(w, "enc_p.mrte." + n, device)
        self.rnd = rnd
        self.cross_attention = _Attention(w, "enc_p.mrte.cross_attention.", device, 4, None, rnd)
        self.c_pre = (rnd(g("c_pre.weight")), g("c_pre.bias"))
        self.text_pre = (rnd(g("text_pre.weight")), g("text_pre.bias"))
        self.c_post = (rnd(g("c_post.weight")), g("c_post.bias"))

    def __call__(self, ssl_enc, ssl_mask, text, text_mask, ge, slice_indices=None):
        if ge is None:
            ge = 0
        if slice_indices is None:
            attn_mask = text_mask.unsqueeze(2) * ssl_mask.unsqueeze(-1)
        else:  # restrict each frame's cross-attention to its own utterance's phonemes (+ the last column)
            rng = torch.arange(text.shape[-1], device=text.device).unsqueeze(0)
            attn_mask = (rng >= slice_indices[:, 0:1]) & (rng < slice_indices[:, 1:2])
            attn_mask[:, -1] = True
            attn_mask = attn_mask.unsqueeze(0).unsqueeze(0)
        rnd = self.rnd
        ssl_enc = rnd(F.conv1d(ssl_enc * ssl_mask, *self.c_pre))
        text_enc = rnd(F.conv1d(text * text_mask, *self.text_pre))
        x = rnd(rnd(self.cross_attention(ssl_enc * ssl_mask, text_enc * text_mask, attn_mask)) + ssl_enc + ge)
        return rnd(F.conv1d(x * ssl_mask, *self.c_post))


class TextEncoder:
    def __init__(self, hps_model, weights, device, round_fn=None):
        m = hps_model
        self.device = device
        rnd = self.rnd = round_fn if round_fn is not None else _ident
        g = lambda n: _dev(weights, "enc_p." + n, device)
        self.out_channels = m["inter_channels"]
        self.ssl_proj = (rnd(g("ssl_proj.weight")), g("ssl_proj.bias"))
        nh, nl, k = m["n_heads"], m["n_layers"], m["kernel_size"]
        self.encoder_ssl = _Encoder(weights, "enc_p.encoder_ssl.", device, nh, nl // 2, k, rnd=rnd)
        self.encoder_text = _Encoder(weights, "enc_p.encoder_text.", device, nh, nl, k, rnd=rnd)
        self.text_embedding = g("text_embedding.weight")
        self.mrte = _MRTE(weights, device, rnd)
        self.encoder2 = _Encoder(weights, "enc_p.encoder2.", device, nh, nl // 2, k, rnd=rnd)
        self.proj = (rnd(g("proj.weight")), g("proj.bias"))
        self.y_overlap = None
        if m["version"] in ("v2Pro", "v2ProPlus"):
            self._ge512 = (_dev(weights, "ge_to512.weight", device), _dev(weights, "ge_to512.bias", device))

    def ge_to512(self, ge):
        """models.py:394: Linear over the channel axis of ge [1, gin, Tg]"""
        return F.linear(ge.transpose(2, 1), *self._ge512).transpose(2, 1)

    def infer(self, y, text, ge, speed, stream_mode=False, valid_start_idx=None, overlap_len=None, slice_indices=None):
        rnd = self.rnd
        y = rnd(y.to(torch.float32))
        y_mask = torch.ones((1, 1, y.size(2)), dtype=y.dtype, device=y.device)
        y = rnd(F.conv1d(y * y_mask, *self.ssl_proj) * y_mask)
        y = self.encoder_ssl(y * y_mask, y_mask)
        text_mask = torch.ones((1, 1, text.size(1)), dtype=y.dtype, device=y.device)
        t = rnd(F.embedding(text, self.text_embedding).transpose(1, 2))
        t = self.encoder_text(t * text_mask, text_mask)
        y = self.mrte(y, y_mask, t, text_mask, ge, slice_indices)
        y = self.encoder2(y * y_mask, y_mask)
        if stream_mode:
            y = y[:, :, valid_start_idx:]
            y_mask = y_mask[:, :, valid_start_idx:]
            alpha = torch.linspace(0, 1, overlap_len, dtype=y.dtype, device=y.device).view(1, 1, -1)
            if self.y_overlap is not None:
                y[:, :, :overlap_len] = self.y_overlap * (1 - alpha) + y[:, :, :overlap_len] * alpha
            self.y_overlap = y[:, :, -overlap_len:]
        if speed != 1:
            y = F.interpolate(y, size=int(y.shape[-1] / speed) + 1, mode="linear")
            y_mask = F.interpolate(y_mask, size=y.shape[-1], mode="nearest")
        stats = F.conv1d(y, *self.proj) * y_mask
        m, logs = torch.split(stats, self.out_channels, dim=1)
        return m, logs, y_mask


class DecodeRestatement:
    """SynthesizerTrn.decode (SoVITS/models.py:385-429) around a `flow_dec(z_p, y_mask, ge)` callable (the product's HIP
    flow + Generator, pinned separately by vocoder.npz): what tests compare the one-call gsv_voc_decode with -- speed != 1
    (features resampled BEFORE proj, :217-219), the streaming slice + cross-fade on the features (:209-215), per-token ge."""

    def __init__(self, hps_model, weights, device, flow_dec, round_fn=None):
        self.w = weights
        self.enc_p = TextEncoder(hps_model, weights, device, round_fn=round_fn)
        self.flow_dec = flow_dec
        self.is_v2pro = hps_model["version"] in ("v2Pro", "v2ProPlus")
        self.device = device

    @torch.inference_mode()
    def __call__(self, codes, text, ge, noise=None, noise_scale=0.0, speed=1, stream_mode=False, valid_start_idx=None,
                 overlap_len=None, slice_indices=None):
        ge = ge.to(device=self.device, dtype=torch.float32)
        if ge.shape[-1] != 1:
            ge = F.interpolate(ge, size=ge.shape[-1] * 2, mode="nearest")
        ge_in = self.enc_p.ge_to512(ge) if self.is_v2pro else ge
        q = codebook_decode(self.w, codes.to(self.device))
        q = F.interpolate(q, size=q.shape[-1] * 2, mode="nearest")
        m_p, logs_p, y_mask = self.enc_p.infer(q, text.to(self.device), ge_in, speed, stream_mode, valid_start_idx, overlap_len, slice_indices)
        if speed != 1 and ge.shape[-1] != 1:
            ge = F.interpolate(ge, size=m_p.shape[-1], mode="nearest")
        z_p = m_p if not noise_scale else m_p + noise.to(m_p) * torch.exp(logs_p) * noise_scale
        return self.flow_dec(z_p, y_mask, ge), self.enc_p.mrte.cross_attention.attn[0], (m_p, logs_p)
