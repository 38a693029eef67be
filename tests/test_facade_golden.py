"""The facade's own arithmetic against tests/golden/facade.npz, which oracle/gen_golden.py
produced by EXECUTING the reference's statements: TTS._find_head/_tail_threshold_offsets (TTS.py:1629-1662), the sort +
both-ends interleave (TTS.py:705-720) and the split / trim loop (TTS.py:806-816) of TTS.infer_batched.  CPU only."""
import os
import sys

import numpy as np
import torch

from oracle.gen_golden_inputs import FACADE_AUDIO, FACADE_LENGTHS, FACADE_SPLITS, SOLA_CASES, facade_audio, sola_case  # noqa: E402

from gsv_tts_lite_amd.batchmath import balance_order, split_bounds  # noqa: E402
from gsv_tts_lite_amd.tts import TTS  # noqa: E402


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "facade.npz"))


def test_head_and_tail_trim_offsets(golden_dir):
    g = _g(golden_dir)
    t = TTS.__new__(TTS)
    for k, case in enumerate(FACADE_AUDIO):
        a = torch.from_numpy(facade_audio(*case))
        assert t._find_head_threshold_offsets(a) == int(g["head"][k]), (k, case)
        assert t._find_tail_threshold_offsets(a) == int(g["tail"][k]), (k, case)


def test_balance_order_is_the_references_sort_and_interleave(golden_dir):
    g = _g(golden_dir)
    for k, lens in enumerate(FACADE_LENGTHS):
        order = balance_order(torch.tensor(lens))
        assert order.tolist() == g["order_%d" % k].tolist(), (k, lens)
        assert (torch.arange(100, 100 + len(lens))[order]).tolist() == g["orig_%d" % k].tolist()


def test_split_and_trim_of_a_time_concatenated_batch(golden_dir):
    g = _g(golden_dir)
    t = TTS.__new__(TTS)
    for k, (lens, speed) in enumerate(FACADE_SPLITS):
        total = int(sum(lens) * 2 * 640 / speed) + 1
        audio = torch.from_numpy(facade_audio(20 + k, total, 0, 0, 0.4))
        pos = 0
        for i, l in enumerate(lens):
            audio[int(pos): int(pos) + 700 * (i + 1)] = 0
            pos += l * 2 * 640 / speed
        sizes, sums = [], []
        for lo, hi in split_bounds(lens, 640, speed):
            a = audio[lo:hi]
            h, tl = t._find_head_threshold_offsets(a), t._find_tail_threshold_offsets(a)
            a = a[h:-tl].float().numpy()
            sizes.append(len(a)); sums.append(float(np.abs(a).astype(np.float64).sum()))
        assert sizes == g["split_%d_sizes" % k].tolist(), (k, sizes)
        np.testing.assert_allclose(sums, g["split_%d_sums" % k], rtol=1e-12)


def test_oracle_sola_is_the_references(golden_dir):
    """oracle.sola against tests/golden/sola.npz (the reference's TTS._sola_algorithm executed on the same seeded cases): the
    offset exactly, the spliced chunk to fp32 rounding of the cross-fade."""
    from oracle import oracle as orc
    g = np.load(os.path.join(golden_dir, "sola.npz"))
    for k, case in enumerate(SOLA_CASES):
        f1, f2 = sola_case(*case)
        out, off = orc.sola(f1, f2, case[2], case[3])
        assert off == int(g["offset_%d" % k]), (k, off)
        ref = g["out_%d" % k]
        assert out.shape == ref.shape and np.abs(out - ref).max() <= 1.2e-7, (k, np.abs(out - ref).max())
