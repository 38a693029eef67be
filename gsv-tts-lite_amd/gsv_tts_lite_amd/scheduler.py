"""Utterance-level data parallelism over the GPUs of one node.

The reference is single-process; its unit of independent work is the utterance (after
`cut_text`, gsv_tts/TTS.py:616-620) and its own continuous-batching loop already treats
utterances as a queue feeding slots (t2s_model.py:696-722).  Scale-out is therefore:
one process per GPU, weights replicated, utterances dealt to ranks, NO collective in the
compute path.  The only exchange is the reference-speaker material produced once per new
speaker/prompt on one rank (`ge` [1,gin,1], prompt tokens, phones1, bert1): one RCCL broadcast
over xGMI (`torch.distributed` backend "nccl" on ROCm), KB-sized, latency-bound.  Results are
variable-length audio and return through the host.

Everything here is backend-agnostic torch.distributed so the N>1 logic is covered on CPU with
gloo (tests/test_scheduler_gloo.py).
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_indices(lengths: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Deal utterances to ranks: sort by expected cost (length) descending and deal round-robin
    in a snake order, so every rank gets the same count (+-1) and near-equal total length.
    Returns the indices (into `lengths`) owned by `rank`, in ascending original order."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    mine = []
    for pos, idx in enumerate(order):
        rnd, off = divmod(pos, world_size)
        owner = off if rnd % 2 == 0 else world_size - 1 - off
        if owner == rank:
            mine.append(idx)
    return sorted(mine)


def broadcast_speaker(tensors: List[torch.Tensor], src: int = 0, group=None) -> List[torch.Tensor]:
    """Broadcast the cached reference-speaker tensors (ge, prompt tokens, phones1, bert1) from the
    rank that ran the reference-audio models.  Shapes must already agree on every rank (they are
    functions of the prompt, known to all ranks); contents are overwritten on non-src ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    for t in tensors:
        dist.broadcast(t, src=src, group=group)
    return tensors


def gather_objects(local, dst: int = 0, group=None):
    """Host-side gather of per-rank results (lists of (orig_index, payload)); returns the merged,
    original-order list on `dst`, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [p for _, p in sorted(local, key=lambda t: t[0])]
    world = dist.get_world_size(group)
    out = [None] * world if dist.get_rank(group) == dst else None
    dist.gather_object(local, out, dst=dst, group=group)
    if out is None:
        return None
    merged = [item for part in out for item in part]
    return [p for _, p in sorted(merged, key=lambda t: t[0])]


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    from .engine import _dist_on          # one rank goes through the backend too under GSV_FORCE_COLLECTIVES (engine.py)
    if not _dist_on(group):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
