"""GPU parity of the subtitle alignment (csrc/align.h through gsv_align_viterbi): bit-exact integer paths
against the reference's outputs (tests/golden/align.npz) and against the oracle on further shapes."""
import os

import numpy as np
import pytest
import torch

from gsv_tts_lite_amd import subtitles as sub
from gsv_tts_lite_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_alignment_golden(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "align.npz"))
    for seed, H, T, P, lead, tail in g["cases"].tolist():
        a = synth.synth_attn(seed, H, T, P, lead, tail)
        got = sub.viterbi_monotonic(torch.from_numpy(a).to(dev)).cpu().numpy()
        assert got.dtype == np.int64
        assert np.array_equal(got, g["assign_%d" % seed]), (seed, H, T, P)


@pytest.mark.parametrize("H,T,P,lead,tail", [
    (4, 2, 7, 0, 0), (4, 9, 7, 1, 1), (4, 10, 64, 0, 2), (4, 16, 65, 0, 0), (4, 17, 256, 2, 3), (4, 18, 257, 0, 0),
    (1, 123, 31, 4, 17), (8, 1000, 200, 7, 40),
    (4, 2500, 130, 3, 50),      # back-pointer bits still in LDS (2500 x 4 words)
    (4, 5000, 260, 3, 50),      # bits spill to the global workspace; 1024-thread walk
    (4, 3000, 1500, 3, 50),     # two phonemes per thread
    (4, 700, 4096, 0, 9),       # four per thread, the ABI's maximum
])
def test_alignment_vs_oracle(dev, H, T, P, lead, tail):
    a = synth.synth_attn(7 * T + P, H, T, P, lead, tail)
    got = sub.viterbi_monotonic(torch.from_numpy(a).to(dev)).cpu().numpy()
    assert np.array_equal(got, orc.viterbi_monotonic(a))


def test_alignment_of_enc_p_attention(dev):
    """the attention the device enc_p produces (softmax rows, incl. the time-concatenated batch form) aligns the
    same on device and in the oracle"""
    rng = np.random.default_rng(11)
    for T, P in [(50, 30), (300, 100)]:
        logits = rng.standard_normal((4, T, P)).astype(np.float32) * 3
        a = torch.softmax(torch.from_numpy(logits), -1).numpy()
        got = sub.viterbi_monotonic(torch.from_numpy(a).to(dev)).cpu().numpy()
        assert np.array_equal(got, orc.viterbi_monotonic(a))


def test_alignment_argument_errors(dev):
    with pytest.raises(RuntimeError):
        sub.viterbi_monotonic(torch.zeros(4, 5, 1, device=dev))       # N < 2
    with pytest.raises(RuntimeError):
        sub.viterbi_monotonic(torch.zeros(4, 5, 5000, device=dev))    # N > 4096
    with pytest.raises(RuntimeError):
        sub.viterbi_monotonic(torch.zeros(9, 5, 50, device=dev))      # more heads than the kernel keeps
