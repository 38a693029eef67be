"""Subtitles: frame->phoneme alignment on the device (csrc/align.h) and the host bookkeeping that turns it into
word timings and source-text spans.

Mirrors the reference's TTS._viterbi_monotonic / _is_normal_assign / _get_subtitles / _find_subtitles /
_cat_subtitles / _increment_subtitle_* (gsv_tts/TTS.py:1664-1808) and TextProcessor.sub2text_index with its
helpers split_text / LIS_mapping / linear_interpolate (gsv_tts/TextProcessor.py:127-233).  The alignment has no
CPU path here: it needs the HIP library."""
import bisect
import re

import torch

from . import _native as N

_WS = {}


def viterbi_monotonic(attn: torch.Tensor) -> torch.Tensor:
    """attn [H, T, N] (cross-attention of enc_p.mrte) -> int64 [T] phoneme per frame, -1 before speech starts
    (TTS.py:1744-1797).  One call of gsv_align_viterbi on the current stream."""
    if attn.dim() != 3:
        raise ValueError("attn must be [heads, frames, phonemes]")
    if not attn.is_cuda:
        raise RuntimeError("viterbi_monotonic runs on the HIP device only (no CPU fallback)")
    L = N.lib()
    a = attn.to(torch.float32).contiguous()
    H, T, P = a.shape
    need = L.gsv_align_workspace(T, P)
    if need == 0:
        raise RuntimeError("gsv_align_workspace: unsupported shape T=%d N=%d" % (T, P))
    ws = _WS.get(a.device)
    if ws is None or ws.numel() < need:
        ws = _WS[a.device] = torch.empty(need, dtype=torch.uint8, device=a.device)
    out = torch.empty(T, dtype=torch.int32, device=a.device)
    N.check(L.gsv_align_viterbi(a.data_ptr(), H, T, P, out.data_ptr(), ws.data_ptr(), ws.numel(), N.current_stream_ptr(a.device)))
    return out.to(torch.int64)


def is_normal_assign(assign, threshold=0.5) -> bool:
    """TTS.py:1799-1808: fewer than `threshold` of the phoneme runs may be single frames."""
    x = [int(v) for v in (assign.tolist() if hasattr(assign, "tolist") else assign) if int(v) != -1]
    if not x:
        return False
    runs, single, n = 0, 0, 1
    for i in range(1, len(x) + 1):
        if i < len(x) and x[i] == x[i - 1]:
            n += 1
        else:
            runs += 1
            single += n == 1
            n = 1
    return single / runs < threshold


def get_subtitles(word2ph, assign, speed, last_end_s=0, sovits_hz=50):
    """TTS.py:1664-1707: word timings from the frame->phoneme path.  A word ends where its last phoneme's run
    of frames ends; a leading -1 run (frames before speech) shifts the first start."""
    a = [int(v) for v in (assign.tolist() if hasattr(assign, "tolist") else assign)]
    frame_time = (1 / sovits_hz) / speed
    run_end_s = [f * frame_time for f in range(1, len(a)) if a[f] != a[f - 1]]
    run_end_s.append(len(a) * frame_time)
    end_s = last_end_s + run_end_s.pop(0) if a[0] == -1 else last_end_s
    out, k = [], -1
    for word, n_ph in zip(word2ph["word"], word2ph["ph"]):
        k += n_ph
        if k >= len(run_end_s):
            break
        start_s, end_s = end_s, run_end_s[k] + last_end_s
        out.append({"text": word, "start_s": start_s, "end_s": end_s})
    if end_s - last_end_s != run_end_s[-1]:
        out.append({"text": word, "start_s": end_s, "end_s": run_end_s[-1] + last_end_s})
    return out


def find_subtitles(subtitles, word2ph, last_i):
    """TTS.py:1709-1719: end index of the run of subtitles that spells this segment's words."""
    w = len(word2ph["word"])
    target = " ".join(word2ph["word"])
    for i in range(last_i, len(subtitles) - w + 1):
        if " ".join(s["text"] for s in subtitles[i:i + w]) == target:
            return i + w
    return len(subtitles)


def cat_subtitles(*subtitle_lists):
    """TTS.py:1721-1731: concatenate per-segment subtitle lists on one time axis."""
    out, last_end = [], 0
    for subs in subtitle_lists:
        shift = subs[0]["start_s"] - last_end
        for s in subs:
            s["start_s"] -= shift
            s["end_s"] -= shift
            out.append(s)
        last_end = subs[-1]["end_s"]
    return out


def increment_subtitle_indices(subtitles, inc):
    for s in subtitles:
        s["orig_idx_start"] += inc
        s["orig_idx_end"] += inc


def increment_subtitle_times(subtitles, inc):
    for s in subtitles:
        s["start_s"] += inc
        if s["end_s"]:
            s["end_s"] += inc


# ---------------------------------------------------------------- normalised text -> source text spans
_TOKEN = re.compile(r"[a-zA-Z]+|.", flags=re.DOTALL)


def split_text(text):
    """TextProcessor.py:127-129: latin words stay whole, everything else is one character."""
    return _TOKEN.findall(text)


def lis_mapping(candidates):
    """TextProcessor.py:131-168: candidates[i] = source positions token i could map to; pick one per token (or
    -1) so that the picked positions form a longest strictly increasing chain."""
    tails = []                      # tails[k] = smallest last position of a chain of length k+1
    seen = []                       # per token: (position, chain length ending there)
    for cand in candidates:
        ranks = [bisect.bisect_left(tails, v) for v in cand]   # all against the tails BEFORE this token
        seen.append([(v, r + 1) for v, r in zip(cand, ranks)])
        for v, r in zip(cand, ranks):
            if r < len(tails):
                tails[r] = min(tails[r], v)
            else:
                tails.append(v)
    out = [-1] * len(candidates)
    want, bound = len(tails), float("inf")
    if want == 0:
        return out
    for i in range(len(candidates) - 1, -1, -1):
        for v, _ in sorted((e for e in seen[i] if e[1] == want), key=lambda e: e[0], reverse=True):
            if v < bound:
                out[i], bound, want = v, v, want - 1
                break
    return out


def linear_interpolate(indices):
    """TextProcessor.py:170-201: fill the -1 holes of a position map linearly (count up after the last hit)."""
    out = list(indices)
    known = [(i, v) for i, v in enumerate(out) if v != -1]
    if not known:
        return out
    i0, v0 = known[0]
    for i in range(i0):
        out[i] = int(round(0 + ((v0 - 0) / i0) * i))
    for (ia, va), (ib, vb) in zip(known, known[1:]):
        for i in range(1, ib - ia):
            out[ia + i] = int(round(va + ((vb - va) / (ib - ia)) * i))
    il, vl = known[-1]
    for i in range(il + 1, len(out)):
        out[i] = vl + (i - il)
    return out


def sub2text_index(subtitles, norm_text, orig_text):
    """TextProcessor.py:203-233: attach [orig_idx_start, orig_idx_end) spans of the caller's text."""
    spans, at = [], 0
    for s in subtitles:
        at = norm_text.find(s["text"], at)
        spans.append((at, at + len(s["text"]) - 1))
    orig_tok, norm_tok = split_text(orig_text), split_text(norm_text)
    picked = lis_mapping([[i for i, t in enumerate(orig_tok) if t == n] for n in norm_tok])
    char_map = []
    for tok, k in zip(norm_tok, picked):
        if k == -1:
            char_map += [-1] * len(tok)
        else:
            base = sum(len(t) for t in orig_tok[:k])
            char_map += list(range(base, base + len(tok)))
    char_map = linear_interpolate(char_map)
    for s, (a, b) in zip(subtitles, spans):
        s["orig_idx_start"] = char_map[a]
        s["orig_idx_end"] = char_map[b] + 1
    return subtitles
