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
