"""Streaming splice on the device: what TTS.infer_stream does between the vocoder and the hand-out of a chunk.

The reference (gsv_tts/TTS.py:429-436) aligns each new chunk to the previous chunk's tail with `_sola_algorithm` (TTS.py:1612-1627:
normalised cross-correlation over 320 candidate offsets + a linear cross-fade, three torch ops and an `.item()`), keeps the last
`overlap` samples back, and hands out the rest.  Here that is one library call per chunk (`gsv_sola`, csrc/sola.h: a score launch
with one block per offset and a splice launch) driven by `ChunkSplicer`, which owns the tail between chunks.  There is no torch
compute on this path and no CPU fallback."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _native as N


def sola(prev_tail: torch.Tensor, chunk: torch.Tensor, search_len: int = 320) -> Tuple[torch.Tensor, int]:
    """prev_tail fp32 [overlap], chunk fp32 [n] (device) -> (spliced chunk fp32 [n - offset], offset)"""
    if not chunk.is_cuda:
        raise RuntimeError("the streaming splice runs on the HIP device only (no CPU fallback)")
    L = N.lib()
    tail = prev_tail.to(device=chunk.device, dtype=torch.float32).reshape(-1).contiguous()
    x = chunk.to(torch.float32).reshape(-1).contiguous()
    n, ov = int(x.numel()), int(tail.numel())
    ws = torch.empty(L.gsv_sola_workspace(int(search_len)), dtype=torch.uint8, device=x.device)
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    off = torch.empty(1, dtype=torch.int32, device=x.device)
    N.check(L.gsv_sola(tail.data_ptr(), x.data_ptr(), n, ov, int(search_len), out.data_ptr(), off.data_ptr(), ws.data_ptr(), ws.numel(),
                       N.current_stream_ptr(x.device)))
    k = int(off.item())          # the chunk's length depends on it; the samples go to the host right after anyway
    return out[:n - k], k


class ChunkSplicer:
    """One streamed utterance: push() every vocoded chunk (which starts `overlap` samples before the previous one ended), get back
    the samples to hand out.  All but the final chunk keep their last `overlap` samples back as the next splice's tail."""

    def __init__(self, overlap_samples: int, search_len: int = 320):
        self.overlap, self.search_len = int(overlap_samples), int(search_len)
        self.tail: Optional[torch.Tensor] = None
        self.offsets = []          # the offset chosen for every spliced chunk (diagnostics, tests)

    def push(self, chunk: torch.Tensor, is_final: bool) -> torch.Tensor:
        x = chunk.reshape(-1)
        if self.tail is not None:
            x, k = sola(self.tail, x, self.search_len)
            self.offsets.append(k)
        self.tail = x[-self.overlap:].clone()
        return x if is_final else x[:-self.overlap]
