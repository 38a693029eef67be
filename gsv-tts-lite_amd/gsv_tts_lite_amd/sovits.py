"""SynthesizerTrn: host-side mirror of the reference's SoVITS runtime for the decode path.

Mirrors gsv_tts/GPT_SoVITS/SoVITS/models.py (class SynthesizerTrn): `initialize_runtime`,
`decode(codes, text, ge, noise_scale, speed, cuda_graph, stream_mode, valid_start_idx,
overlap_len, slice_indices) -> (o, attn)`, `flow_dec(z_p, y_mask, ge) -> o`, and the attributes
TTS reads (`samples_per_frame`, `enc_p.y_overlap`, `enc_p.mrte.cross_attention.attn`).

flow + Generator (models.py:58-65, 113-132 -- the >90% of vocoder time)
run as hand-written HIP behind `gsv_voc_flow_dec`.  `decode()` is ONE C-ABI call (`gsv_voc_decode`): ge_to512, the quantizer
lookup and x2 upsampling, the text/ssl encoder `enc_p` (csrc/encp.h: bf16 MFMA kernels, plain fp32 ones for the parity
mode), the streaming cross-fade, the speed resampling, the noise draw and flow + Generator all run inside the library; torch
only allocates the tensors.  The torch restatement of `enc_p` that the tests compare the kernels with lives with the other
checkers in oracle/sovits_encoder.py.
"""
from __future__ import annotations

import ctypes
import os
import math

import numpy as np
import torch

from . import _native as N

V2PRO_SET = {"v2Pro", "v2ProPlus"}


class _VocoderNative:
    """flow + dec behind the C ABI; owns the native handle and a caller-side workspace."""

    def __init__(self, hps_model, weights, dtype, device):
        m = hps_model
        L = N.lib()
        cfg = N.VocConfig()
        cfg.inter_channels = m["inter_channels"]
        cfg.hidden_channels = m["hidden_channels"]
        cfg.gin_channels = m["gin_channels"]
        cfg.upsample_initial_channel = m["upsample_initial_channel"]
        cfg.n_upsample = len(m["upsample_rates"])
        for i, (u, k) in enumerate(zip(m["upsample_rates"], m["upsample_kernel_sizes"])):
            cfg.upsample_rates[i] = u
            cfg.upsample_kernel_sizes[i] = k
        cfg.n_resblock_kernels = len(m["resblock_kernel_sizes"])
        for i, k in enumerate(m["resblock_kernel_sizes"]):
            cfg.resblock_kernel_sizes[i] = k
        dil = m["resblock_dilation_sizes"][0]
        for d in m["resblock_dilation_sizes"]:
            if list(d) != list(dil):
                raise ValueError("per-resblock dilation sets are not supported")
        for i, d in enumerate(dil):
            cfg.resblock_dilations[i] = d
        cfg.n_flows = 4
        cfg.dtype = N.dtype_code(dtype)
        self.device = torch.device(device)
        self.gin = m["gin_channels"]
        self.inter = m["inter_channels"]
        self.samples_per_frame = int(np.prod(m["upsample_rates"]))
        h = ctypes.c_void_p()
        N.check(L.gsv_voc_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        stream = N.current_stream_ptr(self.device)
        for name, t in weights.items():
            enc = name.startswith("enc_p.") or name.startswith("ge_to512.") or name == "quantizer.vq.layers.0._codebook.embed"
            if not (name.startswith("dec.") or name.startswith("flow.") or enc):
                continue
            d = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            N.check(L.gsv_voc_load_tensor(h, name.encode(), d.data_ptr(), d.numel(), stream))
        N.check(L.gsv_voc_finalize(h, stream))
        self._ws = None
        self._ews = None
        self.has_enc_p = bool(L.gsv_voc_has_enc_p(h))

    def __del__(self):
        try:
            if self._h is not None:
                N.lib().gsv_voc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _workspace(self, T):
        need = N.lib().gsv_voc_workspace(self._h, T)
        if need == 0:
            raise RuntimeError("gsv_voc_workspace failed")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def flow_dec_bucket(self, z_p, y_mask, ge):
        """flow_dec replayed from the hipGraph of this length's bucket (gsv_voc_flow_dec_graph): static buffers are
        allocated once per (T, Tg), the chunk is copied in, the captured pass replayed, the result cloned out -- the
        protocol of the reference's graph path (SoVITS/models.py:406-423)."""
        z, ge, T, Tg = self._prep(z_p, ge)
        key = (T, Tg)
        if not hasattr(self, "_buckets"):
            self._buckets = {}
        b = self._buckets.pop(key, None)
        if b is None:
            while len(self._buckets) >= self.GRAPH_BUCKETS:          # least recently replayed bucket goes (its graph is evicted by
                self._buckets.pop(next(iter(self._buckets)))         # the library when its own cache fills: gsv_voc_flow_dec_graph)
            need = N.lib().gsv_voc_workspace(self._h, T)
            b = {"z": torch.zeros(1, self.inter, T, dtype=torch.float32, device=self.device),
                 "m": torch.zeros(T, dtype=torch.float32, device=self.device),
                 "g": torch.zeros(1, self.gin, Tg, dtype=torch.float32, device=self.device),
                 "o": torch.zeros(1, 1, T * self.samples_per_frame, dtype=torch.float32, device=self.device),
                 "w": torch.empty(need, dtype=torch.uint8, device=self.device)}
        self._buckets[key] = b                                       # most recently used last
        b["z"].copy_(z)
        b["m"].copy_(y_mask.to(device=self.device, dtype=torch.float32).reshape(-1))
        b["g"].copy_(ge)
        N.check(N.lib().gsv_voc_flow_dec_graph(self._h, b["z"].data_ptr(), b["m"].data_ptr(), b["g"].data_ptr(), T, Tg,
                                               b["o"].data_ptr(), b["w"].data_ptr(), b["w"].numel(),
                                               N.current_stream_ptr(self.device)))
        return b["o"].clone()

    def resample_linear(self, x, T_out):
        """[1, C, T] fp32 -> [1, C, T_out], F.interpolate(mode="linear") on the device (gsv_voc_resample_linear)"""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        y = torch.empty(1, x.shape[1], T_out, dtype=torch.float32, device=self.device)
        N.check(N.lib().gsv_voc_resample_linear(x.data_ptr(), x.shape[1], x.shape[2], y.data_ptr(), T_out,
                                                N.current_stream_ptr(self.device)))
        return y

    def _ready(self, t):
        """fp32, contiguous, on the device: what the C ABI takes.  The common case (it already is) costs one check, not two tensor ops --
        a pass that starts on an idle stream pays every host microsecond in front of its first launch (tools/voc_gap.py)."""
        if t.dtype is torch.float32 and t.device == self.device and t.is_contiguous():
            return t
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def _prep(self, z, ge):
        z = self._ready(z)
        ge = self._ready(ge)
        assert z.dim() == 3 and z.shape[0] == 1, "flow/dec run one (possibly time-concatenated) sequence"
        T = z.shape[2]
        Tg = ge.shape[2]
        return z, ge, T, Tg

    # flow_dec(graph replay): measured on MI355X (profiles/r04_vocoder_graph_replay_negative.txt), replaying the ~40-kernel pass of one
    # utterance from a hipGraph is NOT faster for a pass that starts on an idle stream -- hipGraphLaunch spends ~20 us per kernel
    # node before the first kernel runs (bench: 1.19 ms eager -> 1.91 ms replayed), while eager launches (3-4 us each) stay ahead
    # of 5-60 us kernels after the first one.  Replay only pays when passes are queued back to back (streaming chunks:
    # SynthesizerTrn's cuda_graph_buckets).  `auto_graph` therefore defaults to off; it promotes a length to a bucket on its
    # third use when switched on.
    GRAPH_MAX_FRAMES = 1024
    GRAPH_AFTER_USES = 3
    GRAPH_BUCKETS = 8

    def _graph_worthy(self, T, Tg):
        if T > self.GRAPH_MAX_FRAMES or not self.auto_graph:
            return False
        seen = self.__dict__.setdefault("_len_uses", {})
        n = seen.pop((T, Tg), 0) + 1
        seen[(T, Tg)] = n                      # most recently used last
        while len(seen) > 64:
            seen.pop(next(iter(seen)))
        return n >= self.GRAPH_AFTER_USES

    auto_graph = False

    def flow_dec(self, z_p, y_mask, ge):
        if self._graph_worthy(int(z_p.shape[2]), int(ge.shape[2])):
            return self.flow_dec_bucket(z_p, y_mask, ge)
        z, ge, T, Tg = self._prep(z_p, ge)
        mask = self._ready(y_mask).reshape(-1)
        out = torch.empty(1, 1, T * self.samples_per_frame, dtype=torch.float32, device=self.device)
        ws = self._workspace(T)
        N.check(N.lib().gsv_voc_flow_dec(self._h, z.data_ptr(), mask.data_ptr(), ge.data_ptr(), T, Tg, out.data_ptr(),
                                         ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device)))
        return out

    def flow(self, z_p, y_mask, ge):
        z, ge, T, Tg = self._prep(z_p, ge)
        mask = y_mask.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
        out = torch.empty_like(z)
        ws = self._workspace(T)
        N.check(N.lib().gsv_voc_flow(self._h, z.data_ptr(), mask.data_ptr(), ge.data_ptr(), T, Tg, out.data_ptr(),
                                     ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device)))
        return out

    def enc_p(self, codes, text, ge512, slice_indices=None):
        """codes int64 [N], text int64 [P], ge512 fp32 [1, 512, Tg] -> m_p, logs_p [1, inter, 2N], attn [4, 2N, P]"""
        L = N.lib()
        codes = codes.to(device=self.device, dtype=torch.int64).contiguous()
        text = text.to(device=self.device, dtype=torch.int64).contiguous()
        n, P = codes.numel(), text.numel()
        T = 2 * n
        g = ge512.to(device=self.device, dtype=torch.float32)[0].transpose(0, 1).contiguous()   # [Tg][512]
        Tg = g.shape[0]
        sl = None if slice_indices is None else slice_indices.to(device=self.device, dtype=torch.int64).contiguous()
        m_p = torch.empty(1, self.inter, T, dtype=torch.float32, device=self.device)
        logs_p = torch.empty_like(m_p)
        attn = torch.empty(4, T, P, dtype=torch.float32, device=self.device)
        need = L.gsv_voc_enc_workspace(self._h, n, P)
        if need == 0:
            raise RuntimeError("gsv_voc_enc_workspace failed")
        if self._ews is None or self._ews.numel() < need:
            self._ews = torch.empty(need, dtype=torch.uint8, device=self.device)
        N.check(L.gsv_voc_enc_p(self._h, codes.data_ptr(), n, text.data_ptr(), P, g.data_ptr(), Tg,
                                0 if sl is None else sl.data_ptr(), m_p.data_ptr(), logs_p.data_ptr(), attn.data_ptr(),
                                self._ews.data_ptr(), self._ews.numel(), N.current_stream_ptr(self.device)))
        return m_p, logs_p, attn

    def decode(self, codes, text, ge, slice_indices, noise_scale, seed, T_out, valid_start, overlap_len, overlap_state,
               has_overlap, bucket):
        """gsv_voc_decode: codes int64 [n], text int64 [P], ge fp32 [gin, Tg] (Tg = 1 or n) -> (audio [1, 1, T_out * hop],
        attn [4, 2n, P]).  `T_out`: the frame count after the speed change, evaluated ONCE by the caller (models.py:217) -- it
        sizes `out` here and is what the library resamples to.  `bucket`: replay flow + Generator from the hipGraph of this
        chunk length (its own workspace)."""
        L = N.lib()
        n, P, Tg = int(codes.numel()), int(text.numel()), int(ge.shape[-1])
        T_out = int(T_out)
        need = L.gsv_voc_decode_workspace(self._h, n, P, Tg, T_out, int(valid_start))
        if need == 0:
            raise RuntimeError("gsv_voc_decode_workspace failed (n_codes %d, n_text %d, Tg %d, out_frames %d, valid_start %d)" % (n, P, Tg, T_out, valid_start))
        if bucket:
            if not hasattr(self, "_dws_bucket"):
                self._dws_bucket = {}
            key = (n, Tg, valid_start, T_out)
            ws = self._dws_bucket.pop(key, None)
            if ws is None or ws.numel() < need:     # sized for longer texts than this one: a regrown workspace is a re-captured graph
                ws = torch.empty(max(need, L.gsv_voc_decode_workspace(self._h, n, max(P, 384), Tg, T_out, int(valid_start))),
                                 dtype=torch.uint8, device=self.device)
            self._dws_bucket[key] = ws              # most recently used last; the library evicts graphs the same way
            while len(self._dws_bucket) > 32:
                self._dws_bucket.pop(next(iter(self._dws_bucket)))
        else:
            if getattr(self, "_dws", None) is None or self._dws.numel() < need:
                self._dws = torch.empty(need, dtype=torch.uint8, device=self.device)
            ws = self._dws
        out = torch.empty(1, 1, T_out * self.samples_per_frame, dtype=torch.float32, device=self.device)
        attn = torch.empty(4, 2 * n, P, dtype=torch.float32, device=self.device)
        N.check(L.gsv_voc_decode(self._h, codes.data_ptr(), n, text.data_ptr(), P, ge.data_ptr(), Tg,
                                 0 if slice_indices is None else slice_indices.data_ptr(), float(noise_scale), int(seed) & (2 ** 64 - 1),
                                 T_out, int(valid_start), int(overlap_len), 0 if overlap_state is None else overlap_state.data_ptr(),
                                 1 if has_overlap else 0, 1 if bucket else 0, out.data_ptr(), attn.data_ptr(), ws.data_ptr(), ws.numel(),
                                 N.current_stream_ptr(self.device)))
        return out, attn

    def dec(self, z, ge):
        z, ge, T, Tg = self._prep(z, ge)
        out = torch.empty(1, 1, T * self.samples_per_frame, dtype=torch.float32, device=self.device)
        ws = self._workspace(T)
        N.check(N.lib().gsv_voc_dec(self._h, z.data_ptr(), ge.data_ptr(), T, Tg, out.data_ptr(), ws.data_ptr(),
                                    ws.numel(), N.current_stream_ptr(self.device)))
        return out


class _Holder:
    pass


class _EncPState:
    """what TTS reads / resets on `vq_model.enc_p` (TTS.py:498, models.py:427): the streaming cross-fade state and the MRTE
    attention of the last decode().  The encoder itself lives in the library (csrc/encp.h)."""

    def __init__(self):
        self.y_overlap = None        # fp32 [2 * inter, overlap_len]: the projected statistics of the previous chunk's tail
        self.mrte = _Holder()
        self.mrte.cross_attention = _Holder()
        self.mrte.cross_attention.attn = None


class SynthesizerTrn:
    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, n_speakers=0, gin_channels=0,
                 semantic_frame_rate="25hz", freeze_quantizer=None, version="v2", **kwargs):
        self.hps_model = dict(inter_channels=inter_channels, hidden_channels=hidden_channels,
                              filter_channels=filter_channels, n_heads=n_heads, n_layers=n_layers,
                              kernel_size=kernel_size, p_dropout=p_dropout, resblock=resblock,
                              resblock_kernel_sizes=list(resblock_kernel_sizes),
                              resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes],
                              upsample_rates=list(upsample_rates), upsample_initial_channel=upsample_initial_channel,
                              upsample_kernel_sizes=list(upsample_kernel_sizes), gin_channels=gin_channels,
                              version=version)
        self.inter_channels = inter_channels
        self.gin_channels = gin_channels
        self.upsample_rates = list(upsample_rates)
        self.samples_per_frame = math.prod(self.upsample_rates)
        self.semantic_frame_rate = semantic_frame_rate
        self.version = version
        self.is_v2pro = version in V2PRO_SET
        self.cuda_graph_buckets = []
        self._weights = None
        self._voc = None
        self.enc_p = None
        self._ref = None

    def load_state_dict(self, sd, strict=False):
        self._weights = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
                         for k, v in sd.items()}

    def eval(self):
        return self

    @torch.inference_mode()
    def initialize_runtime(self, dtype, device, sovits_caches):
        """models.py:322-369.  The reference captures one CUDA graph per cache length.  Here a pass whose length EQUALS a
        bucket (the streaming chunk sizes, default [50, 55]) replays a hipGraph of the whole flow + Generator pass
        (gsv_voc_flow_dec_graph, captured on first use); other lengths run eagerly -- the reference pads them to the
        next bucket, which lets the padded frames' conditioning leak into the last real frames, so
        the eager pass is the faithful one for them."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the MI355X hot path needs a GPU device; there is no CPU fallback")
        self.device, self.dtype = device, dtype
        self._voc = _VocoderNative(self.hps_model, self._weights, dtype, device)
        self.cuda_graph_buckets = sorted(sovits_caches)
        # hot-path-only weight sets (no enc_p tensors): flow_dec still works, decode() raises
        self.enc_p = _EncPState() if self._voc.has_enc_p else None

    def _ref_audio(self):
        if self._ref is None:
            from .refaudio import RefAudioNative, has_ref_tensors
            if self._voc is None:
                raise RuntimeError("call initialize_runtime first")
            if not has_ref_tensors(self._weights):
                raise RuntimeError("get_ge / extract_latent need the ref_enc.* / ssl_proj.* tensors in the state dict")
            self._ref = RefAudioNative(self._weights, self.gin_channels, self.is_v2pro, self.device)
        return self._ref

    @torch.inference_mode()
    def spectrogram(self, audio):
        """the Spectrogram transform of TTS._get_spec (TTS.py:1591-1604) for this model's filter / hop length"""
        return self._ref_audio().spectrogram(audio)

    @torch.inference_mode()
    def get_ge(self, refer, sv_emb=None):
        """models.py:371-378, on the device (csrc/refaudio.h)"""
        return self._ref_audio().get_ge(refer, sv_emb if self.is_v2pro else None)

    @torch.inference_mode()
    def extract_latent(self, x):
        """models.py:431-434, on the device"""
        return self._ref_audio().extract_latent(x)

    def flow_dec(self, z_p, y_mask, ge):
        """models.py:380-383"""
        return self._voc.flow_dec(z_p, y_mask, ge)

    @torch.inference_mode()
    def decode(self, codes, text, ge, noise_scale=0.5, speed=1, cuda_graph=True, stream_mode=False,
               valid_start_idx=None, overlap_len=None, slice_indices=None, generator=None):
        """models.py:385-429, one library call (gsv_voc_decode).  codes int64 [1, 1, N], text int64 [1, P], ge [1, gin, 1] or --
        a time-concatenated batch -- [1, gin, N] (one column per TOKEN, what TTS.infer_batched builds; the reference
        interpolates it to 2N frames, here that is an index map).  The noise is the library's counter-based stream, seeded from
        `generator` (or torch's default CPU generator), not torch.randn's."""
        if self.enc_p is None:
            raise RuntimeError("decode() needs the enc_p / quantizer tensors in the state dict")
        if codes.dim() != 3 or codes.shape[0] != 1 or codes.shape[1] != 1:
            raise ValueError("decode() takes codes of shape [1, 1, N] (one quantizer, batch of one): the reference never passes "
                             "anything else, and a time-concatenated batch is still a batch of one")
        dev = self.device
        n = int(codes.shape[-1])
        codes = codes.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        text = text.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        ge = ge.to(device=dev, dtype=torch.float32).reshape(self.gin_channels, -1).contiguous()
        if ge.shape[-1] not in (1, n):
            raise ValueError("ge must have 1 column or one per token (%d), got %d" % (n, ge.shape[-1]))
        sl = None if slice_indices is None else slice_indices.to(device=dev, dtype=torch.int64).contiguous()
        seed = 0
        if noise_scale != 0:
            # the 64-bit seed of this call's noise stream is DRAWN from the generator (torch's default CPU generator without one),
            # as the reference's torch.randn_like draws from it (models.py:404): the generator's state advances per call, and
            # re-seeding it replays the same sequence of calls
            gdev = torch.device("cpu") if generator is None else generator.device
            seed = int(torch.empty((), dtype=torch.int64, device=gdev).random_(generator=generator).item()) & (2 ** 64 - 1)
        start, ov, state, has = 0, 0, None, False
        if stream_mode:
            start, ov = int(valid_start_idx), int(overlap_len)
            state, has = self.enc_p.y_overlap, self.enc_p.y_overlap is not None
            if state is None or tuple(state.shape) != (2 * self.inter_channels, ov):
                state, has = torch.zeros(2 * self.inter_channels, ov, dtype=torch.float32, device=dev), False
            self.enc_p.y_overlap = state          # updated in place by the call: the tail of this chunk's statistics
        T_out = 2 * n - start if speed == 1 else int((2 * n - start) / speed) + 1
        bucket = bool(cuda_graph) and T_out in self.cuda_graph_buckets and ge.shape[-1] == 1
        o, attn = self._voc.decode(codes, text, ge, sl, noise_scale, seed, T_out, start, ov, state, has, bucket)
        self.enc_p.mrte.cross_attention.attn = attn[None]
        return o, attn
