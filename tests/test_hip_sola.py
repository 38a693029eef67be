"""GPU parity of the streaming splice (csrc/sola.h behind gsv_sola) against tests/golden/sola.npz -- the reference's
TTS._sola_algorithm (gsv_tts/TTS.py:1612-1627) executed on the seeded cases of oracle/gen_golden_inputs.SOLA_CASES -- and against
the numpy restatement (oracle.sola) on fresh inputs; plus the chunk protocol of TTS.infer_stream through ChunkSplicer."""
import os

import numpy as np
import pytest
import torch

from oracle.gen_golden_inputs import SOLA_CASES, facade_audio, sola_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_sola_equals_the_reference_golden(dev, golden_dir):
    from gsv_tts_lite_amd.stream import sola
    g = np.load(os.path.join(golden_dir, "sola.npz"))
    for k, case in enumerate(SOLA_CASES):
        f1, f2 = sola_case(*case)
        out, off = sola(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev), case[3])
        ref = g["out_%d" % k]
        assert off == int(g["offset_%d" % k]), (k, off, int(g["offset_%d" % k]))          # bit-exact index
        got = out.cpu().numpy()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        e = np.abs(got - ref).max() if got.size else 0.0
        assert e <= 1.2e-7, (k, e)                                                         # one fp32 rounding of the cross-fade
        assert np.array_equal(got[case[2]:], ref[case[2]:])                                # behind the fade: copied samples


def test_sola_vs_oracle_on_fresh_inputs_and_short_chunks(dev):
    from oracle import oracle as orc
    from gsv_tts_lite_amd.stream import sola
    rng = np.random.default_rng(11)
    for n, ov, search in [(5000, 640, 320), (700, 640, 320), (640, 640, 320), (4000, 1, 50), (12800, 3200, 0), (9000, 3200, 1000)]:
        base = rng.standard_normal(n + 2 * search + ov + 8).astype(np.float32)
        base = np.convolve(base, np.ones(7, np.float32) / 7, mode="same").astype(np.float32)
        shift = int(rng.integers(0, max(1, min(search, max(0, n - ov)) + 1)))
        at = search + 4
        f1 = base[at: at + ov].copy()
        f2 = (base[at - shift: at - shift + n] + 0.05 * rng.standard_normal(n)).astype(np.float32)
        want, woff = orc.sola(f1, f2, ov, search)
        got, off = sola(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev), search)
        assert off == woff, (n, ov, search, off, woff, shift)
        if ov >= 64:           # a one-sample overlap scores every candidate +-|tail|: the FIRST maximum wins, not the planted shift
            assert off == shift, (n, ov, search, off, shift)
        assert got.shape[0] == want.shape[0] and np.abs(got.cpu().numpy() - want).max() <= 1.2e-7


def test_sola_rejects_bad_arguments(dev):
    from gsv_tts_lite_amd import _native as N
    from gsv_tts_lite_amd.stream import sola
    with pytest.raises(RuntimeError):
        sola(torch.zeros(100, device=dev), torch.zeros(50, device=dev))          # chunk shorter than the overlap
    with pytest.raises(RuntimeError):
        sola(torch.zeros(10), torch.zeros(50))                                   # host tensors: no CPU path
    assert N.lib().gsv_sola_workspace(-1) == 0


def test_chunk_splicer_protocol(dev):
    """TTS.py:429-436: every chunk but the last keeps `overlap` samples back; the next chunk is aligned to that tail.  A stream cut
    into overlapping chunks of one long signal, each delayed by a known shift, reassembles into that signal."""
    from gsv_tts_lite_amd.stream import ChunkSplicer
    ov, n_chunk = 640, 6400
    sig = facade_audio(300, 40000, 0, 0, 0.5)
    sm = np.convolve(sig, np.ones(5, np.float32) / 5, mode="same").astype(np.float32)
    sp = ChunkSplicer(ov)
    out, pos, shifts = [], 0, [0, 13, 200, 0, 77]
    for c, sh in enumerate(shifts):
        final = c == len(shifts) - 1
        start = pos - sh if c else 0                           # the vocoder re-renders the held-back tail, `sh` samples late
        chunk = torch.from_numpy(sm[start: start + n_chunk].copy()).to(dev)
        piece = sp.push(chunk[None, None], final)
        out.append(piece.cpu().numpy())
        pos = start + n_chunk if final else start + n_chunk - ov     # end of what has been handed out
    got = np.concatenate(out)
    assert sp.offsets == shifts[1:]
    assert got.shape[0] == pos and np.abs(got - sm[:pos]).max() < 1e-6
