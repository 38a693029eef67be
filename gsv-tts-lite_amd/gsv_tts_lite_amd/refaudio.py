"""Reference-audio path on the device: the spectrogram of TTS._get_spec, the speaker
embedding `ge` of SynthesizerTrn.get_ge, and the prompt tokens of SynthesizerTrn.extract_latent -- csrc/refaudio.h
behind the gsv_ref_* entry points.  Runs in fp32 whatever the handle's numerics mode: it happens once per new
speaker / prompt and feeds everything after it.  No CPU path: it needs the HIP library."""
import ctypes

import torch

from . import _native as N

REF_PREFIXES = ("ref_enc.", "sv_emb.", "ssl_proj.")
REF_NAMES = ("prelu.weight", "quantizer.vq.layers.0._codebook.embed")


def has_ref_tensors(weights) -> bool:
    return "ref_enc.fc.fc.weight" in weights and "ssl_proj.weight" in weights


class RefAudioNative:
    def __init__(self, weights, gin, is_v2pro, device, n_fft=2048, hop=640):
        L = N.lib()
        cfg = N.RefConfig(n_fft=n_fft, hop=hop, spec_bins=704, hidden=128, n_head=2, kernel=5, gin=gin,
                          sv_dim=20480 if (is_v2pro and "sv_emb.weight" in weights) else 0, ssl_dim=768, bins=1024)
        self.device = torch.device(device)
        self.gin, self.n_fft, self.hop, self.has_sv = gin, n_fft, hop, cfg.sv_dim > 0
        h = ctypes.c_void_p()
        N.check(L.gsv_ref_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        st = N.current_stream_ptr(self.device)
        for name, t in weights.items():
            if not (name.startswith(REF_PREFIXES) or name in REF_NAMES):
                continue
            if name.startswith("sv_emb.") or name == "prelu.weight":
                if not self.has_sv:
                    continue
            d = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            N.check(L.gsv_ref_load_tensor(h, name.encode(), d.data_ptr(), d.numel(), st))
        N.check(L.gsv_ref_finalize(h, st))
        self._ws = None

    def __del__(self):
        try:
            if self._h is not None:
                N.lib().gsv_ref_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _workspace(self, n_samples=0, n_frames=0, n_ssl=0):
        need = N.lib().gsv_ref_workspace(self._h, n_samples, n_frames, n_ssl)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        return self._ws

    def spectrogram(self, audio):
        """audio [n] or [1, n] fp32 at the model rate -> [1, n_fft/2+1, 1 + n//hop] (TTS.py:1591-1604)"""
        a = audio.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
        n = a.numel()
        spec = torch.empty(1, self.n_fft // 2 + 1, 1 + n // self.hop, dtype=torch.float32, device=self.device)
        ws = self._workspace(n_samples=n)
        N.check(N.lib().gsv_ref_spectrogram(self._h, a.data_ptr(), n, spec.data_ptr(), ws.data_ptr(), ws.numel(),
                                            N.current_stream_ptr(self.device)))
        return spec

    def get_ge(self, refer, sv_emb=None):
        """refer [1, bins >= 704, T], sv_emb [1, 20480] or None -> ge [1, gin, 1] (models.py:371-378)"""
        r = refer.to(device=self.device, dtype=torch.float32)
        if r.dim() != 3 or r.shape[0] != 1 or r.shape[1] < 704:
            raise ValueError("refer must be [1, >=704, T]")
        r = r[0].contiguous()
        T = r.shape[1]
        sv = None
        if sv_emb is not None and self.has_sv:
            sv = sv_emb.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
            if sv.numel() != 20480:
                raise ValueError("sv_emb must have 20480 values")
        ge = torch.empty(1, self.gin, 1, dtype=torch.float32, device=self.device)
        ws = self._workspace(n_frames=T)
        N.check(N.lib().gsv_ref_get_ge(self._h, r.data_ptr(), T, 0 if sv is None else sv.data_ptr(), ge.data_ptr(),
                                       ws.data_ptr(), ws.numel(), N.current_stream_ptr(self.device)))
        return ge

    def extract_latent(self, ssl, return_margin=False):
        """ssl [1, 768, Th] -> codes int64 [1, 1, Th // 2] (models.py:431-434)"""
        x = ssl.to(device=self.device, dtype=torch.float32)
        if x.dim() != 3 or x.shape[0] != 1 or x.shape[1] != 768:
            raise ValueError("ssl must be [1, 768, Th]")
        x = x[0].contiguous()
        Th = x.shape[1]
        codes = torch.empty(1, 1, Th // 2, dtype=torch.int64, device=self.device)
        margin = torch.empty(Th // 2, dtype=torch.float32, device=self.device)
        ws = self._workspace(n_ssl=Th)
        N.check(N.lib().gsv_ref_extract_latent(self._h, x.data_ptr(), Th, codes.data_ptr(), margin.data_ptr(), ws.data_ptr(),
                                               ws.numel(), N.current_stream_ptr(self.device)))
        return (codes, margin) if return_margin else codes
