"""Index arithmetic of TTS.infer_batched around the vocoder batches, as functions so that it can be pinned against the
reference's own statements (tests/golden/facade.npz is produced by executing gsv_tts/TTS.py:705-720 and :806-816)."""
from __future__ import annotations

import torch


def balance_order(lengths: torch.Tensor) -> torch.Tensor:
    """TTS.py:705-716: sort the utterances by token count, then interleave the sorted list from both ends
    (shortest, longest, 2nd shortest, 2nd longest, ...) so that consecutive vocoder batches carry similar totals.
    Returns the permutation to apply to the completion-order lists."""
    order = torch.argsort(lengths)
    n = len(order)
    inter = torch.zeros(n, dtype=torch.long, device=lengths.device)
    srt = torch.arange(n, device=lengths.device)
    inter[0::2] = srt[:(n + 1) // 2]
    inter[1::2] = srt[(n + 1) // 2:].flip(0)
    return order[inter]


def split_bounds(lengths, samples_per_frame: int, speed: float):
    """TTS.py:806-811: sample ranges of the utterances inside one time-concatenated vocoder batch.  The running end is
    a float (lengths * 2 * samples_per_frame / speed accumulates un-rounded); each slice is [int(start), int(end))."""
    out, pos = [], 0
    for l in lengths:
        nxt = pos + l * 2 * samples_per_frame / speed
        out.append((int(pos), int(nxt)))
        pos = nxt
    return out
