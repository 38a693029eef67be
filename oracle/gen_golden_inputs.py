"""TEST INFRASTRUCTURE: seeded INPUTS of tests/golden/facade.npz, shared by the generator (oracle/gen_golden.py, build
container only) and tests/test_facade_golden.py (anywhere).  No reference code, no outputs -- inputs only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gsv-tts-lite_amd"))
from gsv_tts_lite_amd import synth  # noqa: E402

FACADE_AUDIO = [  # (seed, n_samples, lead silence, tail silence, gain)
    (0, 96000, 9000, 12000, 0.5), (1, 40000, 0, 0, 0.8), (2, 64000, 70000, 0, 0.5), (3, 900, 100, 0, 0.5),
    (4, 150000, 30000, 90000, 0.3), (5, 20000, 5000, 4000, 0.015), (6, 512, 0, 0, 0.9), (7, 70000, 1000, 69000, 0.6),
]
FACADE_LENGTHS = [[5], [7, 3], [4, 9, 2], [10, 10, 10, 10], [3, 8, 1, 9, 4, 7, 2, 6, 5, 12, 11], list(range(17, 0, -1)),
                  [6, 2, 6, 2, 6, 1, 1, 9]]


def facade_audio(seed, n, lead, tail, gain):
    a = (synth.hashed_uniform("facade.audio.%d" % seed, (n,), 77) * np.float32(2.0) - np.float32(1.0)) * np.float32(gain)
    a[:min(lead, n)] = 0
    if tail:
        a[max(0, n - tail):] = 0
    return a.astype(np.float32)

FACADE_SPLITS = [([3, 5, 2], 1.0), ([4, 1, 6, 2], 1.3), ([7], 0.8), ([2, 2, 2, 2, 2], 1.1)]


# streaming splice (TTS._sola_algorithm): (seed, chunk samples, overlap samples, search_len, true shift of the chunk against the tail,
# gain of the uncorrelated noise added to the chunk).  The chunk repeats the signal the tail was cut from, `shift` samples late.
SOLA_CASES = [(0, 35200, 3200, 320, 137, 0.05), (1, 16000, 3200, 320, 0, 0.1), (2, 9000, 640, 320, 320, 0.02), (3, 3400, 3200, 320, 55, 0.05),
              (4, 3200, 3200, 320, 0, 0.05), (5, 6400, 1280, 64, 17, 0.3), (6, 8000, 3200, 320, -1, 0.0)]


def sola_case(seed, n, overlap, search, shift, noise):
    """-> (f1_overlap [overlap], f2 [n]) float32; shift < 0: an all-zero chunk against an all-zero tail (every score ties at 0)"""
    if shift < 0:
        return np.zeros(overlap, np.float32), np.zeros(n, np.float32)
    base = facade_audio(100 + seed, n + 2 * search + overlap + 8, 0, 0, 0.6)
    # low-pass a little (audio is not white): a 5-tap box, in float32, fixed order
    sm = base.copy()
    for d in (1, 2):
        sm[d:] += base[:-d]
        sm[:-d] += base[d:]
    sm = (sm * np.float32(0.2)).astype(np.float32)
    at = search + 4
    f1 = sm[at: at + overlap].copy()
    f2 = sm[at - shift: at - shift + n].copy() * np.float32(0.9)
    f2 = f2 + facade_audio(200 + seed, n, 0, 0, noise) if noise else f2
    return f1.astype(np.float32), f2.astype(np.float32)
