"""GPU parity tests of the SoVITS flow + Generator (HIP, through the C ABI) vs the reference's
golden outputs and the CPU oracle.  fp32 mode: north_star tolerance 1e-3 abs on the waveform
(measured ~5e-6); bf16 mode: bounded error (bf16 activations/weights, fp32 accumulate)."""
import os

import numpy as np
import pytest
import torch

from gsv_tts_lite_amd import synth

pytestmark = pytest.mark.gpu
CASES = [("v2Pro", 50, "c"), ("v2Pro", 55, "pf"), ("v2ProPlus", 50, "c"), ("v2", 23, "c"), ("v2Pro", 200, "c"), ("v2ProPlus", 55, "pf")]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _voc(ver, seed, dtype, dev):
    from gsv_tts_lite_amd.sovits import _VocoderNative
    hps = synth.sovits_hps(ver)
    w = synth.sovits_weights(hps, seed=seed, hot_path_only=True)
    return _VocoderNative(hps["model"], {k: torch.from_numpy(a) for k, a in w.items()}, dtype, dev), hps, w


def _T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("ver,T,tag", CASES)
def test_flow_dec_fp32_matches_reference_golden(golden_dir, dev, ver, T, tag):
    g = np.load(os.path.join(golden_dir, "vocoder.npz"))
    v, _, _ = _voc(ver, int(g["seed"]), torch.float32, dev)
    name = "%s_T%d_%s" % (ver, T, tag)
    z, ge = _T(g[name + "_z"], dev), _T(g[name + "_ge"], dev)
    mask = torch.ones(1, 1, T, device=dev)
    zf = v.flow(z, mask, ge)
    np.testing.assert_allclose(zf.cpu().numpy(), g[name + "_flow"], atol=1e-4)
    od = v.dec(_T(g[name + "_flow"], dev), ge)
    sub = name + "_o" not in g            # long cases store every 5th sample
    want = g[name + "_o_sub"] if sub else g[name + "_o"]
    pick = (lambda a: a[::5]) if sub else (lambda a: a)
    np.testing.assert_allclose(pick(od.cpu().numpy()[0, 0]), want, atol=1e-4)
    o = v.flow_dec(z, mask, ge)
    assert o.shape == (1, 1, T * 640)
    np.testing.assert_allclose(pick(o.cpu().numpy()[0, 0]), want, atol=1e-3)       # north_star bound
    assert np.abs(pick(o.cpu().numpy()[0, 0]) - want).max() < 1e-4                 # what we actually hold


def test_flow_dec_fp32_vs_oracle_odd_lengths_and_mask(dev):
    """lengths that are not tile multiples, a ragged tail masked to zero, per-frame ge."""
    from oracle import oracle as orc
    v, hps, w = _voc("v2Pro", 9, torch.float32, dev)
    vo = orc.VocoderOracle(hps, w)
    for T, per_frame in [(1, False), (7, False), (131, True)]:
        z = synth.hashed_uniform("odd.z%d" % T, (1, 192, T), 9) * np.float32(1.3)
        mask = np.ones(T, np.float32)
        if T > 20:
            mask[-9:] = 0.0
        ge = synth.synth_ge(3, 1024, 9)
        if per_frame:
            ge = np.concatenate([np.repeat(synth.synth_ge(i, 1024, 9), n, axis=2) for i, n in ((3, 40), (4, T - 40))], axis=2)
        ref = vo.flow_dec(z[0], mask, ge[0])
        out = v.flow_dec(_T(z, dev), _T(mask, dev).reshape(1, 1, T), _T(ge, dev))[0, 0].cpu().numpy()
        assert np.abs(out - ref).max() < 1e-4, (T, np.abs(out - ref).max())


def test_generator_linearity_property_full_size(dev):
    """size-independent property at the benchmark size (T=500): the Generator's first conv and
    conditioning are affine in z, so dec(z) must be invariant to how z is split in time only via
    its receptive field -- check causality of the halo: samples far from a perturbed frame are
    unchanged (receptive field of the stack is finite), and the output is finite and bounded."""
    v, _, _ = _voc("v2Pro", 1234, torch.float32, dev)
    T = 500
    z = _T(synth.hashed_uniform("prop.z", (1, 192, T), 1) * np.float32(1.2), dev)
    ge = _T(synth.synth_ge(0, 1024, 1), dev)
    o1 = v.dec(z, ge)
    z2 = z.clone(); z2[:, :, 400] += 1.0
    o2 = v.dec(z2, ge)
    assert torch.isfinite(o1).all() and o1.abs().max() <= 1.0
    d = (o1 - o2).abs()[0, 0]
    assert d[: 300 * 640].max() == 0.0, "perturbing frame 400 changed samples 100 frames away"
    assert d[395 * 640: 405 * 640].max() > 0


@pytest.mark.parametrize("ver,T,tag", CASES[:3])
def test_flow_dec_bf16_bounded(golden_dir, dev, ver, T, tag):
    g = np.load(os.path.join(golden_dir, "vocoder.npz"))
    v, _, _ = _voc(ver, int(g["seed"]), torch.bfloat16, dev)
    name = "%s_T%d_%s" % (ver, T, tag)
    o = v.flow_dec(_T(g[name + "_z"], dev), torch.ones(1, 1, T, device=dev), _T(g[name + "_ge"], dev))[0, 0].cpu().numpy()
    ref = g[name + "_o"]
    err = np.abs(o - ref)
    assert np.isfinite(o).all()
    assert err.max() < 8e-2 and err.mean() < 8e-3, (err.max(), err.mean())


def test_flow_bf16_fused_vs_oracle_ragged(dev):
    """The bf16 flow runs the fused coupling-layer kernel (csrc/flowfuse.h): one tile, several tiles,
    lengths that are not multiples of the 48-row tile, a tail masked to zero, broadcast and per-frame
    conditioning -- against the fp32 oracle flow.  Tolerance = bf16 activations (8 mantissa bits)
    through 16 gated layers; masked rows must come back exactly zero."""
    from oracle import oracle as orc
    v, hps, w = _voc("v2Pro", 11, torch.bfloat16, dev)
    vo = orc.VocoderOracle(hps, w)
    for T, per_frame in [(1, False), (37, False), (48, True), (49, False), (131, True), (500, False)]:
        z = synth.hashed_uniform("ff.z%d" % T, (1, 192, T), 11) * np.float32(1.3)
        mask = np.ones(T, np.float32)
        if T > 20:
            mask[T - 9:] = 0
        ge = synth.synth_ge(3, 1024, 11)
        if per_frame:
            ge = np.concatenate([np.repeat(synth.synth_ge(i, 1024, 11), n, axis=2) for i, n in ((3, 20), (4, T - 20))], axis=2)
        ref = vo.flow(z[0], mask, ge[0])
        out = v.flow(_T(z, dev), _T(mask, dev).reshape(1, 1, T), _T(ge, dev))[0].cpu().numpy()
        assert out.shape == ref.shape and np.isfinite(out).all()
        err = np.abs(out - ref)
        assert err.max() < 6e-2 and err.mean() < 6e-3, (T, err.max(), err.mean())
        if T > 20:
            assert np.abs(out[:, T - 9:]).max() == 0.0


@pytest.mark.parametrize("ver,T,tag", CASES[:2])
def test_flow_bf16_fused_matches_reference_golden(golden_dir, dev, ver, T, tag):
    g = np.load(os.path.join(golden_dir, "vocoder.npz"))
    v, _, _ = _voc(ver, int(g["seed"]), torch.bfloat16, dev)
    name = "%s_T%d_%s" % (ver, T, tag)
    zf = v.flow(_T(g[name + "_z"], dev), torch.ones(1, 1, T, device=dev), _T(g[name + "_ge"], dev)).cpu().numpy()
    err = np.abs(zf - g[name + "_flow"])
    assert err.max() < 6e-2 and err.mean() < 6e-3, (err.max(), err.mean())


@pytest.mark.parametrize("ver", ["v2Pro", "v2ProPlus"])
def test_generator_bf16_wconv_odd_lengths_vs_oracle(dev, ver):
    """bf16 Generator (wconv / tapgemm / conv_post kernels) at lengths that leave ragged last tiles in every
    stage (T = 1, 7, 131 -> 10 .. 83840 rows), broadcast and per-frame ge, against the fp32 oracle.  v2Pro walks
    the 256 (K split in the block, slices over 4 blocks) / 128 / 64 / 32 / 16-channel wconv shapes, v2ProPlus the
    192 (3 blocks) / 96 (3 slices + a staging wave) / 48 (half-empty second slice) / 24-in-32 ones."""
    from oracle import oracle as orc
    v, hps, w = _voc(ver, 13, torch.bfloat16, dev)
    vo = orc.VocoderOracle(hps, w)
    for T, per_frame in [(1, False), (7, True), (131, False)]:
        z = synth.hashed_uniform("bfodd.z%d" % T, (1, 192, T), 13) * np.float32(1.2)
        ge = synth.synth_ge(2, 1024, 13)
        if per_frame:
            ge = np.concatenate([np.repeat(synth.synth_ge(i, 1024, 13), n, axis=2) for i, n in ((2, 3), (5, T - 3))], axis=2)
        ref = vo.dec(z[0], ge[0])
        out = v.dec(_T(z, dev), _T(ge, dev))[0, 0].cpu().numpy()
        assert out.shape == ref.shape and np.isfinite(out).all()
        err = np.abs(out - ref)
        assert err.max() < 8e-2 and err.mean() < 8e-3, (T, err.max(), err.mean())


def test_bench_shape_bf16_flow_and_generator_vs_bf16_oracle(dev):
    """The kernels bench.py times (v2Pro, bf16, T = 500 frames = 10 s of audio) against the oracle in ITS bf16 mode:
    bf16 weights and every stored activation rounded where the HIP path stores bf16 (oracle/gsv_oracle.c ORC_R_VOC),
    so only fp32 summation order -- and the one-ulp operand flips it causes, which ~100 conv layers then spread --
    separates the two.  Measured on MI355X: flow max 1.6e-2 / mean 2.0e-3 (vs the fp32 oracle 2.2e-2 / 3.0e-3);
    Generator max 2.5e-2 / mean 3.2e-3 on a waveform of rms 0.5 -- and NOT closer to the rounding-matched oracle than to
    the fp32 one (2.6e-2 / 3.2e-3): after a hundred layers the flips have decorrelated the two bf16 computations as far
    as bf16 is from fp32.  So what this test pins is the bench-size run itself (T = 500; the older bf16 check stopped at
    T = 131 with max < 8e-2), with bounds = measured distance + headroom."""
    from oracle import oracle as orc
    T = 500
    v, hps, w = _voc("v2Pro", 1234, torch.bfloat16, dev)
    o16 = orc.VocoderOracle(hps, w, numerics="bf16")
    o32 = orc.VocoderOracle(hps, w)
    z = synth.hashed_uniform("bench.z", (1, 192, T), 1234) * np.float32(1.2)
    ge = synth.synth_ge(0, 1024, 1234)
    mask = np.ones(T, np.float32)
    zf = v.flow(_T(z, dev), torch.ones(1, 1, T, device=dev), _T(ge, dev))[0].cpu().numpy()
    f16, f32 = o16.flow(z[0], mask, ge[0]), o32.flow(z[0], mask, ge[0])
    e16, e32 = np.abs(zf - f16), np.abs(zf - f32)
    print("flow T=500 bf16: vs bf16 oracle max %.2e mean %.2e | vs fp32 oracle max %.2e mean %.2e" % (e16.max(), e16.mean(), e32.max(), e32.mean()))
    assert e16.max() < 3e-2 and e16.mean() < 4e-3, (e16.max(), e16.mean())
    # Generator alone on the oracle's own flow output (identical inputs), then the whole pass
    out = v.dec(_T(f16[None], dev), _T(ge, dev))[0, 0].cpu().numpy()
    d16, d32 = o16.dec(f16, ge[0]), o32.dec(f16, ge[0])
    e16, e32 = np.abs(out - d16), np.abs(out - d32)
    print("Generator T=500 bf16: vs bf16 oracle max %.2e mean %.2e | vs fp32 oracle max %.2e mean %.2e" % (e16.max(), e16.mean(), e32.max(), e32.mean()))
    assert out.shape == (T * 640,) and np.isfinite(out).all()
    assert e16.max() < 5e-2 and e16.mean() < 5e-3, (e16.max(), e16.mean())
    full = v.flow_dec(_T(z, dev), torch.ones(1, 1, T, device=dev), _T(ge, dev))[0, 0].cpu().numpy()
    ef = np.abs(full - o16.flow_dec(z[0], mask, ge[0]))
    print("flow_dec T=500 bf16 vs bf16 oracle: max %.2e mean %.2e" % (ef.max(), ef.mean()))
    assert ef.max() < 8e-2 and ef.mean() < 8e-3, (ef.max(), ef.mean())


def test_flow_dec_switches_to_graph_replay_on_a_repeated_length(dev):
    """With auto_graph on, _VocoderNative.flow_dec replays a length from a hipGraph from its third use on (sovits.py:
    GRAPH_AFTER_USES; off by default: a replay from an idle stream measured slower than eager launches); the replayed pass must
    return exactly what the eager pass returns, for new inputs too, and long passes must stay eager."""
    from gsv_tts_lite_amd.sovits import _VocoderNative
    hps = synth.sovits_hps("v2Pro")
    w = synth.sovits_weights(hps, seed=3, hot_path_only=True)
    v = _VocoderNative(hps["model"], {k: torch.from_numpy(a) for k, a in w.items()}, torch.bfloat16, dev)
    ge = torch.from_numpy(synth.synth_ge(0, 1024, 3)).to(dev)
    T = 77
    outs = []
    v.auto_graph = True
    for k in range(5):
        z = torch.from_numpy(synth.hashed_uniform("gr.z%d" % k, (1, 192, T), 3)).to(dev)
        o = v.flow_dec(z, torch.ones(1, 1, T, device=dev), ge)
        v.auto_graph = False
        e = v.flow_dec(z, torch.ones(1, 1, T, device=dev), ge)
        v.auto_graph = True
        assert torch.equal(o, e), k
        outs.append(o)
    assert (T, 1) in v._buckets and len(v._buckets) == 1
    assert not torch.equal(outs[3], outs[4])
    for _ in range(4):       # a long (batched) pass never takes a bucket
        v.flow_dec(torch.zeros(1, 192, v.GRAPH_MAX_FRAMES + 8, device=dev), torch.ones(1, 1, v.GRAPH_MAX_FRAMES + 8, device=dev), ge)
    assert len(v._buckets) == 1


def test_per_frame_ge_runs_on_its_distinct_columns(dev):
    """A time-concatenated batch hands flow_dec one ge column per frame; the library finds the distinct columns on the device and
    runs the conditioning GEMMs on those only (csrc/voc_kernels.h: seg_*).  The result must equal, bit for bit, the pass that
    conditions every frame (GSV_NO_GE_SEGMENTS=1) -- for one speaker, for several utterances of different speakers (also with a
    speaker coming back), for columns that all differ, at a length that takes the staged flow and one that takes the fused kernel."""
    import os
    from gsv_tts_lite_amd.sovits import _VocoderNative
    hps = synth.sovits_hps("v2Pro")
    w = synth.sovits_weights(hps, seed=11, hot_path_only=True)
    v = _VocoderNative(hps["model"], {k: torch.from_numpy(a) for k, a in w.items()}, torch.bfloat16, dev)
    g = [torch.from_numpy(synth.synth_ge(i, 1024, 11)).to(dev) for i in range(4)]

    def cols(spec):
        return torch.cat([g[i].expand(-1, -1, n) for i, n in spec], 2).contiguous()
    cases = [("one speaker", 300, cols([(0, 300)])),
             ("three utterances", 700, cols([(0, 250), (1, 130), (2, 320)])),
             ("a speaker comes back", 2500, cols([(0, 900), (1, 700), (0, 600), (3, 300)])),
             ("every column differs", 96, torch.randn(1, 1024, 96, device=dev))]
    for name, T, ge in cases:
        z = torch.from_numpy(synth.hashed_uniform("seg.z%d" % T, (1, 192, T), 11)).to(dev)
        m = torch.ones(1, 1, T, device=dev)
        a = v.flow_dec(z, m, ge)
        os.environ["GSV_NO_GE_SEGMENTS"] = "1"
        try:
            b = v.flow_dec(z, m, ge)
        finally:
            del os.environ["GSV_NO_GE_SEGMENTS"]
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b), name
    # one speaker per frame == the same speaker broadcast (the reference's two ways of saying the same thing)
    T = 300
    z = torch.from_numpy(synth.hashed_uniform("seg.z%d" % T, (1, 192, T), 11)).to(dev)
    m = torch.ones(1, 1, T, device=dev)
    assert torch.equal(v.flow_dec(z, m, cols([(0, T)])), v.flow_dec(z, m, g[0]))


def test_wdma_pass_equals_the_wconv_pass_bit_for_bit(dev, tmp_path):
    """csrc/wdma.h (rows and residual by LDS-DMA, the activated copies written by their producers, the swizzled LDS map, the
    register epilogue) against the path it replaced (wconv.h: rows through registers, leaky-ReLU at staging; GSV_NO_WDMA=1):
    the same arithmetic at the same rounding points, so flow + Generator must return the SAME samples -- at a length that is not
    a multiple of any tile (edge tiles, zero page, sink), at the bench length, and for a ten-utterance batch that hands the
    256-channel stage to cgemm.  The switch is read once per process: two child processes.  (Round 6: the shipped library has no
    wconv.h kernel for 256 channels any more -- GSV_NO_WDMA=1 moves the 64 / 128-channel stages and leaves 256 on wdma.h.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os, numpy as np, torch\n"
        "sys.path[:0] = [%r, %r]\n"
        "from gsv_tts_lite_amd import synth\n"
        "from gsv_tts_lite_amd.sovits import _VocoderNative\n"
        "dev = torch.device('cuda:0'); hps = synth.sovits_hps('v2Pro')\n"
        "w = synth.sovits_weights(hps, seed=21, hot_path_only=True)\n"
        "v = _VocoderNative(hps['model'], {k: torch.from_numpy(a) for k, a in w.items()}, torch.bfloat16, dev)\n"
        "ge = torch.from_numpy(synth.synth_ge(2, 1024, 21)).to(dev)\n"
        "out = {}\n"
        "for T in (137, 500, 3301):\n"
        "    z = torch.from_numpy(synth.hashed_uniform('wdma.z%%d' %% T, (1, 192, T), 21)).to(dev)\n"
        "    g = ge if T < 1000 else ge.expand(-1, -1, T).contiguous()\n"
        "    out['t%%d' %% T] = v.flow_dec(z, torch.ones(1, 1, T, device=dev), g).cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n" % (root, os.path.join(root, "gsv-tts-lite_amd")))
    outs = []
    for tag, extra in (("wdma", {}), ("wconv", {"GSV_NO_WDMA": "1"})):
        f = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ); env.pop("GSV_NO_WDMA", None); env.update(extra)
        p = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(np.load(f))
    for k in outs[0].files:
        a, b = outs[0][k], outs[1][k]
        assert np.isfinite(a).all() and a.shape == b.shape and np.abs(a).max() > 1e-3, k
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))


def test_cgemm_k_split_is_deterministic_replayable_and_close_to_the_unsplit_pass(dev, tmp_path):
    """csrc/cgemm.h K split (round 6): with few row tiles (v2ProPlus' 384-channel stage at <= 10 s of audio, its 192-channel stage in a
    streaming chunk) a tile's 64-channel chunks are dealt to consecutive blocks, the partial tiles meet through global memory behind
    flags.  The hand-off must be invisible: (a) repeated passes return the SAME samples (a stale or early-read partial tile would
    not), (b) a hipGraph replay of the pass -- the flags reset themselves, no host code runs between replays -- equals the eager
    pass, replay after replay, (c) the pass with GSV_CGEMM_NO_SPLIT=1 (one block per tile: another fp32 summation order) stays
    within bf16 rounding noise of it."""
    import subprocess
    import sys
    from gsv_tts_lite_amd.sovits import _VocoderNative
    hps = synth.sovits_hps("v2ProPlus")
    w = synth.sovits_weights(hps, seed=23, hot_path_only=True)
    v = _VocoderNative(hps["model"], {k: torch.from_numpy(a) for k, a in w.items()}, torch.bfloat16, dev)
    ge = torch.from_numpy(synth.synth_ge(1, 1024, 23)).to(dev)
    ref = {}
    for T in (9, 131, 500):
        z = torch.from_numpy(synth.hashed_uniform("cgs.z%d" % T, (1, 192, T), 23)).to(dev)
        m = torch.ones(1, 1, T, device=dev)
        first = v.flow_dec(z, m, ge)
        assert torch.isfinite(first).all() and float(first.abs().max()) > 1e-3
        for _ in range(6):
            assert torch.equal(v.flow_dec(z, m, ge), first), T
        for _ in range(3):
            assert torch.equal(v.flow_dec_bucket(z, m, ge), first), T
        ref[T] = first.cpu().numpy()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os, numpy as np, torch\n"
        "sys.path[:0] = [%r, %r]\n"
        "from gsv_tts_lite_amd import synth\n"
        "from gsv_tts_lite_amd.sovits import _VocoderNative\n"
        "dev = torch.device('cuda:0'); hps = synth.sovits_hps('v2ProPlus')\n"
        "w = synth.sovits_weights(hps, seed=23, hot_path_only=True)\n"
        "v = _VocoderNative(hps['model'], {k: torch.from_numpy(a) for k, a in w.items()}, torch.bfloat16, dev)\n"
        "ge = torch.from_numpy(synth.synth_ge(1, 1024, 23)).to(dev)\n"
        "out = {}\n"
        "for T in (9, 131, 500):\n"
        "    z = torch.from_numpy(synth.hashed_uniform('cgs.z%%d' %% T, (1, 192, T), 23)).to(dev)\n"
        "    out['t%%d' %% T] = v.flow_dec(z, torch.ones(1, 1, T, device=dev), ge).cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n" % (root, os.path.join(root, "gsv-tts-lite_amd")))
    f = str(tmp_path / "nosplit.npz")
    env = dict(os.environ); env["GSV_CGEMM_NO_SPLIT"] = "1"
    p = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    un = np.load(f)
    differs = False
    for T in (9, 131, 500):
        a, b = ref[T], un["t%d" % T]
        d = np.abs(a - b)
        assert d.max() < 8e-2 and d.mean() < 8e-3, (T, float(d.max()), float(d.mean()))
        differs = differs or bool(d.max() > 0)
    assert differs, "GSV_CGEMM_NO_SPLIT=1 returned the split pass's samples bit for bit: the switch (or the split) is not taken"
