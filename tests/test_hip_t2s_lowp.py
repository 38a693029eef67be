"""GPU parity of the REDUCED-PRECISION GPT kernels -- the ones bench.py times -- against the oracle in its matching
numerics mode (oracle.T2SOracle(numerics="bf16" | "fp8")): weights, K/V and GEMM operands are rounded at the same
places as in the HIP path (oracle/gsv_oracle.c, ORC_R_*), so only the fp32 summation order separates the two.

What that does and does not allow (measured on MI355X, and the reason for each bound below):
  * a 1e-6 fp32 difference occasionally lands on the other side of a bf16 rounding boundary: single K/V elements or
    GEMM operands then differ by ONE bf16 ulp (2^-8 relative), and every following LayerNorm + GEMM spreads that.
    Measured distance to the bf16 oracle: hidden states max 5e-3 / mean 7e-4 after 3 layers, max 2e-2 / mean 2e-3
    after 24 (the bf16-vs-fp32 distance is 5e-2 / 1e-2); layer-0 K/V rows, which see no upstream flips, are
    compared as "nothing further apart than one ulp";
  * greedy tokens must agree wherever the oracle's top-1 / top-2 logit gap exceeds TOKEN_MARGIN = 5e-2: the measured
    logit noise of those flips after 24 layers.  Rounds 3-4 (the decode step's activations entered its dots with 16
    significant bits): largest gap at an observed divergence 9.4e-3 bf16, 1.7e-2 fp8, gate 2e-2.  Round 5: an activation is ONE
    bf16 operand (the reference's own bf16 path multiplies bf16 activations; csrc/t2s_decode.h kPairAct), i.e. four more
    operand vectors per layer that a 1e-6 difference can flip by a bf16 ulp: the same weak step of the bench request (step 107)
    now shows a gap of 2.6e-2 in the oracle, hence 5e-2 -- still 7x tighter than the bf16-vs-fp32 gate of the fp32-referenced
    tests (0.35).
"""
import numpy as np
import pytest
import torch

from gsv_tts_lite_amd import synth

pytestmark = pytest.mark.gpu

TOKEN_MARGIN = 5e-2      # logits are O(6): gaps above this must give the same argmax
HID_TOL = 1e-3
FP8_MARGIN = 0.6         # an e4m3 operand flip is 16x a bf16 one: measured logit noise up to 0.36 after 24 layers


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _model(cfg, w, cache, dtype, dev):
    from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
    m = Text2SemanticDecoder(cfg)
    m.load_state_dict(w)
    m.initialize_runtime(dtype, dev, cache)
    return m


def _first_mismatch(tok, ref):
    nm = min(len(tok), len(ref))
    neq = np.nonzero(np.asarray(tok[:nm]) != np.asarray(ref[:nm]))[0]
    if neq.size:
        return int(neq[0])
    return None if len(tok) == len(ref) else nm


def _kv_close(got, want):
    """bf16 cache rows of layer 0 (no upstream layers): a flipped bf16 operand of the QKV GEMM moves an output by
    ~1e-4, i.e. by a few ulps where the value is small -- so: nearly all elements within one ulp, none off by 2e-3"""
    d = np.abs(got - want)
    ulp = np.maximum(np.abs(want), 1e-2) * 2.0 ** -7
    assert not (d > np.maximum(ulp * 1.01, 2e-3)).any(), float(d.max())
    assert (d > ulp * 1.01).mean() < 2e-3 and d.mean() < 1e-4, ((d > ulp * 1.01).mean(), d.mean())


@pytest.mark.parametrize("n_layer", [3, 24])
def test_bf16_prefill_and_decode_hidden_vs_bf16_oracle(dev, n_layer):
    """bench prompt shape: prompt pass (rowgemm / MFMA attention) and three per-sequence decode steps.  3 layers: the
    tight bound.  24 layers: one-ulp operand flips (module docstring) are amplified by every following LayerNorm, so the
    bound is what 24 layers of that amplification measure, still 10x below the bf16-vs-fp32 distance (5e-2)."""
    from oracle import oracle as orc
    cfg = synth.gpt_config(n_layer=n_layer)
    w = synth.gpt_weights(cfg, seed=1234)
    x, y, bert, _ = synth.synth_request(3, 40, 60, 100, seed=1234, bert="random")
    cache = [(1, 256)]
    o = orc.T2SOracle(cfg, w, cache, numerics="bf16")
    m = _model(cfg, w, cache, torch.bfloat16, dev)
    xy_ref = np.concatenate([o.embed_text(x, bert), o.embed_audio(y)])[None]
    h_ref = o.prefill(xy_ref, o.single_mask(len(x), len(y))[None], 1, 0)
    xy, xl, yl, _, _ = m.embed_prompt([_T(x, dev)], [_T(y, dev)], [_T(bert, dev)])
    assert np.abs(xy.cpu().numpy() - xy_ref).max() < HID_TOL
    m.prefill(1, 0, xy, xl, yl)
    L = len(x) + len(y)
    err = np.abs(xy.cpu().numpy()[0] - h_ref[0])
    print("prefill hidden vs bf16 oracle, %d layers: max %.2e mean %.2e" % (n_layer, err.max(), err.mean()))
    if n_layer <= 3:
        assert err.max() < 1e-2 and err.mean() < 1.5e-3, (err.max(), err.mean())
    else:
        assert err.max() < 4e-2 and err.mean() < 4e-3, (err.max(), err.mean())
    rt = m._rt[1]
    _kv_close(rt["k"][0, 0, :, :L].float().cpu().numpy(), o.cache[1][0][0, 0, :, :L])   # layer 0: no upstream flips
    _kv_close(rt["v"][0, 0, :, :L].float().cpu().numpy(), o.cache[1][1][0, 0, :, :L])
    for lay in range(1, n_layer):
        for t, oc in ((rt["k"], o.cache[1][0]), (rt["v"], o.cache[1][1])):
            d = np.abs(t[lay, 0, :, :L].float().cpu().numpy() - oc[lay, 0, :, :L])
            assert d.mean() < 5e-3 and d.max() < 1e-1, (lay, d.mean(), d.max())
    rng = np.random.default_rng(0)
    kv = L
    # align the two caches before the decode steps (they differ by the flips above): the decode kernels are then
    # compared on identical K/V
    with torch.inference_mode():
        for t, oc in ((rt["k"], o.cache[1][0]), (rt["v"], o.cache[1][1])):
            t[:, 0, :, :L] = torch.from_numpy(oc[:, 0, :, :L]).to(dev).to(t.dtype)
    for _ in range(3):
        xin = rng.normal(size=(1, 512)).astype(np.float32)
        want = o.decode(xin, 1, [kv])
        got = m.decode_hidden(1, _T(xin, dev)).cpu().numpy()
        kv += 1
        e = np.abs(got - want).max()
        print("decode hidden vs bf16 oracle, %d layers: max %.2e" % (n_layer, e))
        assert e < (HID_TOL if n_layer <= 3 else 10 * HID_TOL), e


def test_bf16_greedy_tokens_bench_shape_vs_bf16_oracle(dev):
    """the bench workload itself: 24 layers, 100 phonemes + 100 prompt tokens, 250 greedy tokens (kv 200 -> 450), bf16,
    hipGraph on.  Every token must equal the bf16 oracle's up to the first step whose oracle margin is below
    TOKEN_MARGIN; the run must not be trivially short."""
    from oracle import oracle as orc
    cfg = synth.gpt_config()
    w = synth.gpt_weights(cfg, seed=1234, eos_gain=0.0)
    cache = [(1, 256), (1, 450)]
    agree = []
    for i in (0, 1):
        x, y, bert, _ = synth.synth_request(i, 40, 60, 100, seed=1234)
        o = orc.T2SOracle(cfg, w, cache, numerics="bf16")
        ref = o.infer(x, y, bert, top_k=1)
        m = _model(cfg, w, cache, torch.bfloat16, dev)
        tok = m.infer(_T(x, dev)[None], _T(y, dev)[None], _T(bert, dev)[None], top_k=1)[0, 0].cpu().numpy()
        assert len(ref) == 250 and len(tok) == 250
        first = _first_mismatch(tok, ref)
        if first is not None:
            print("request %d: first divergence at step %d, oracle top-1/top-2 gap there %.3e (gate %.0e); smallest gap before it %.3e"
                  % (i, first, o.margins[first + 1], TOKEN_MARGIN, min(o.margins[:first + 1])))
            assert o.margins[first + 1] < TOKEN_MARGIN, (i, first, o.margins[first + 1])
        agree.append(250 if first is None else first)
        del m
    print("bf16 tokens equal to the bf16 oracle for the first %s of 250 steps" % agree)
    # BOTH requests: measured 250 and 107 matched steps (r3, r4); the gate keeps 25 % headroom under the shorter one.  The
    # matched prefix is a property of (weights seed, request, summation order): a kernel change that re-orders a bf16 sum may
    # move it, and the margin assertion above is what decides whether such a move is legitimate
    assert min(agree) >= 80 and max(agree) >= 200, agree


@pytest.mark.parametrize("dtype,numerics", [(torch.float32, "fp32"), (torch.bfloat16, "bf16")])
def test_continuous_batching_40_requests_32_slots(dev, dtype, numerics):
    """BASELINE configs[2]'s slot count: 40 mixed-length requests through 32 slots (8 refills), greedy.  fp32 handles run the
    two-sequences-per-block decode kernels at 32 slots (csrc/t2s_decode_multi.h); bf16 handles run the batched MFMA chain from
    17 sequences on (csrc/t2s_small.h, `batched_min`).  fp32: tokens, completion order and semantic_orig_idx bit-exact against
    the oracle's continuous batching.  bf16: each request equal to the bf16 oracle's up to the first step whose margin is below
    TOKEN_MARGIN.  (The full-size 256-request / 24-layer run is tests/test_hip_cb_fullsize.py.)"""
    from oracle import oracle as orc
    cfg = synth.gpt_config(n_layer=4)
    w = synth.gpt_weights(cfg, seed=41, eos_gain=2.0)
    cache = [(32, 160)]
    rng = np.random.default_rng(7)
    shapes = [(int(rng.integers(2, 10)), int(rng.integers(3, 24)), int(rng.integers(4, 40))) for _ in range(40)]
    rs = [synth.synth_request(300 + i, p, t, n, seed=41, bert="random") for i, (p, t, n) in enumerate(shapes)]
    m = _model(cfg, w, cache, dtype, dev)
    o = orc.T2SOracle(cfg, w, cache, numerics=numerics, batched_min=m.batched_min, ffn_slices=m.ffn_slices)
    ref, ref_idx = o.infer_batched([r[0] for r in rs], [r[1] for r in rs], [r[2] for r in rs], top_k=1)
    pred, idx = m.infer_batched([_T(r[0], dev) for r in rs], [_T(r[1], dev) for r in rs], [_T(r[2], dev) for r in rs], top_k=1)
    assert sorted(idx.tolist()) == list(range(40))
    if numerics == "fp32":
        assert idx.tolist() == ref_idx.tolist()
        for a_, b_ in zip(pred, ref):
            assert np.array_equal(a_.cpu().numpy(), b_)
        return
    ref_by_req = {int(i): t for i, t in zip(ref_idx, ref)}
    exact = 0
    for req, tok in zip(idx.tolist(), pred):
        first = _first_mismatch(tok.cpu().numpy(), ref_by_req[req])
        if first is None:
            exact += 1
        elif first + 1 < len(o.req_margins[req]):
            assert o.req_margins[req][first + 1] < TOKEN_MARGIN, (req, first, o.req_margins[req][first + 1])
    print("bf16 batched step: %d of 40 requests token-identical to the bf16 oracle" % exact)
    assert exact >= 30, exact


# 17: the first batch size on the chain (one row in its second row tile); 40: a half-empty tile; 64 / 100 / 256: four channel tiles
# per block (csrc/t2s_small.h), a ragged last tile, the largest batch
@pytest.mark.parametrize("B,n_layer", [(17, 1), (40, 1), (64, 1), (100, 1), (256, 1), (64, 24)])
def test_batched_step_hidden_vs_oracle_bf16_and_fp8(dev, B, n_layer):
    """one decode step of the batched chain (5 launches per layer) on B sequences with ragged cache lengths: final
    hidden states against the oracle in the matching numerics mode.  ONE layer is the arithmetic check (fragment
    layouts, per-channel scales, e4m3 conversion, LayerNorm prologue): no flip has a layer to grow in.  24 layers bound
    the growth: a flipped e4m3 operand is a 2^-4 relative step (bf16: 2^-8), so fp8 drifts 16x further from its oracle."""
    from oracle import oracle as orc
    cfg = synth.gpt_config(n_layer=n_layer)
    w = synth.gpt_weights(cfg, seed=99)
    cache = [(B, 96)]
    rng = np.random.default_rng(B)
    shapes = [(int(rng.integers(2, 8)), int(rng.integers(3, 20)), int(rng.integers(4, 30))) for _ in range(B)]
    rs = [synth.synth_request(500 + i, p, t, n, seed=99, bert="random") for i, (p, t, n) in enumerate(shapes)]
    xin = rng.normal(size=(B, 512)).astype(np.float32)
    o32 = orc.T2SOracle(cfg, w, cache)
    results = {}
    for dtype, numerics in ((torch.bfloat16, "bf16"), (torch.float8_e4m3fn, "fp8")):
        m = _model(cfg, w, cache, dtype, dev)
        if B < m.batched_min:
            pytest.skip("batch below the batched-step threshold")
        o = orc.T2SOracle(cfg, w, cache, numerics=numerics, batched_min=m.batched_min, ffn_slices=m.ffn_slices)
        xy, xl, yl, xlh, ylh = m.embed_prompt([_T(r[0], dev) for r in rs], [_T(r[1], dev) for r in rs], [_T(r[2], dev) for r in rs])
        m.prefill(B, 0, xy, xl, yl)
        kv = (xlh + ylh).numpy()
        for ora in (o, o32):                      # the oracle's cache rows come from ITS prompt pass (same roundings)
            Lm = int(kv.max())
            xyo = np.zeros((B, Lm, 512), np.float32)
            mask = np.zeros((B, Lm, Lm), np.uint8)
            for b, r in enumerate(rs):
                lx, ly = len(r[0]), len(r[1])
                xyo[b, :lx] = ora.embed_text(r[0], r[2]); xyo[b, lx:lx + ly] = ora.embed_audio(r[1])
                mask[b, :lx + ly, :lx + ly] = ora.single_mask(lx, ly)
            ora.prefill(xyo, mask, B, 0)
        rt = m._rt[B]                             # identical K/V on both sides: the step itself is what is compared
        with torch.inference_mode():
            for t, oc in ((rt["k"], o.cache[B][0]), (rt["v"], o.cache[B][1])):
                t.copy_(torch.from_numpy(oc).to(dev).to(t.dtype))
        # gsv_t2s_decode_hidden takes the path gsv_t2s_decode would: the batched chain from batched_min sequences on
        want = o.decode(xin, B, kv)
        want32 = o32.decode(xin, B, kv)
        got = m.decode_hidden(B, _T(xin, dev)).cpu().numpy()
        err = np.abs(got - want)
        results[numerics] = (float(err.max()), float(err.mean()), float(np.abs(got - want32).max()), float(np.abs(got - want32).mean()))
        lim = {("bf16", 1): (3e-3, 3e-4), ("fp8", 1): (5e-2, 3e-3), ("bf16", 24): (4e-2, 4e-3), ("fp8", 24): (0.5, 4e-2)}[(numerics, n_layer)]
        assert err.max() < lim[0] and err.mean() < lim[1], (numerics, n_layer, err.max(), err.mean())
        del m
    print("batched step, B=%d, %d layers: (max, mean) |hidden - oracle(same numerics)|, (max, mean) vs fp32 arithmetic: %s" % (B, n_layer, results))


def test_fp8_batched_tokens_match_rate(dev):
    """BASELINE configs[4]: fp8 QKV / FFN at bs=64.  64 requests, 24 layers, greedy, until the cache is full: equality with the
    fp8-mode oracle gated on its margins, and the match rate against the fp32 reference arithmetic is REPORTED (fp8
    operands legitimately flip close decisions; it must stay well above chance and is printed for the record)."""
    from oracle import oracle as orc
    cfg = synth.gpt_config()
    w = synth.gpt_weights(cfg, seed=77, eos_gain=0.0)
    B, T = 64, 96
    cache = [(B, T)]
    rs = [synth.synth_request(900 + i, 6, 10 + i % 7, 20 + i % 11, seed=77) for i in range(B)]
    m = _model(cfg, w, cache, torch.float8_e4m3fn, dev)
    o8 = orc.T2SOracle(cfg, w, cache, numerics="fp8", batched_min=m.batched_min, ffn_slices=m.ffn_slices)
    o32 = orc.T2SOracle(cfg, w, cache)
    pred, idx = m.infer_batched([_T(r[0], dev) for r in rs], [_T(r[1], dev) for r in rs], [_T(r[2], dev) for r in rs], top_k=1)
    got = {int(i): t.cpu().numpy() for i, t in zip(idx.tolist(), pred)}
    ref8, i8 = o8.infer_batched([r[0] for r in rs], [r[1] for r in rs], [r[2] for r in rs], top_k=1)
    ref32, i32 = o32.infer_batched([r[0] for r in rs], [r[1] for r in rs], [r[2] for r in rs], top_k=1)
    r8 = {int(i): t for i, t in zip(i8, ref8)}
    r32 = {int(i): t for i, t in zip(i32, ref32)}
    same8 = tot = same32 = 0
    for req in range(B):
        first = _first_mismatch(got[req], r8[req])
        n = len(r8[req])
        if first is not None and first + 1 < len(o8.req_margins[req]):
            assert o8.req_margins[req][first + 1] < FP8_MARGIN, (req, first, o8.req_margins[req][first + 1])
        same8 += n if first is None else first
        f32 = _first_mismatch(got[req], r32[req])
        same32 += min(len(got[req]), len(r32[req])) if f32 is None else f32
        tot += n
    print("fp8 bs=64: tokens before the first divergence / total: vs fp8 oracle %.3f, vs fp32 reference arithmetic %.3f"
          % (same8 / tot, same32 / tot))
    assert same8 / tot > 0.25     # every divergence above was at a margin below FP8_MARGIN; this only guards against garbage
    assert same32 / tot > 0.15


def test_ffn_slice_count_is_reported_and_part_of_the_arithmetic(dev):
    """gsv_t2s_ffn_slices: 64 slices of 32 hidden units at <= 4 sequences, 32 of 64 otherwise (fp32 handles take the same switch
    since round 5: their partials stay fp32, only the summation order moves, and the fp32 tokens stay bit-exact against the
    reference: test_hip_t2s.py, test_hip_bench_size.py).  On bf16 handles the slice partials are rounded to half, so the count is part of the arithmetic: one decode step of 4 and of 5
    sequences (either side of the switch) against the bf16-mode oracle summing the SAME slices, and against the other count --
    which must be further away than the matching one on at least one of the two (the check would be vacuous otherwise)."""
    from oracle import oracle as orc
    cfg = synth.gpt_config(n_layer=2)
    w = synth.gpt_weights(cfg, seed=5)
    m32 = _model(cfg, w, [(1, 64)], torch.float32, dev)
    assert [m32.ffn_slices(b) for b in (1, 4, 5, 16)] == [64, 64, 32, 32]
    del m32
    dist = {}
    for B in (4, 5):
        cache = [(B, 64)]
        m = _model(cfg, w, cache, torch.bfloat16, dev)
        assert m.ffn_slices(B) == (64 if B <= 4 else 32) and m.ffn_slices(1) == 64 and m.ffn_slices(16) == 32
        rng = np.random.default_rng(B)
        rs = [synth.synth_request(700 + i, 4, 6 + i, 8 + i, seed=5, bert="random") for i in range(B)]
        xin = rng.normal(size=(B, 512)).astype(np.float32)
        xy, xl, yl, xlh, ylh = m.embed_prompt([_T(r[0], dev) for r in rs], [_T(r[1], dev) for r in rs], [_T(r[2], dev) for r in rs])
        m.prefill(B, 0, xy, xl, yl)
        kv = (xlh + ylh).numpy()
        got = None
        for name, fn in (("same", m.ffn_slices), ("other", lambda b: 96 - m.ffn_slices(b))):
            o = orc.T2SOracle(cfg, w, cache, numerics="bf16", batched_min=m.batched_min, ffn_slices=fn)
            Lm = int(kv.max())
            xyo = np.zeros((B, Lm, 512), np.float32)
            mask = np.zeros((B, Lm, Lm), np.uint8)
            for b, r in enumerate(rs):
                lx, ly = len(r[0]), len(r[1])
                xyo[b, :lx] = o.embed_text(r[0], r[2]); xyo[b, lx:lx + ly] = o.embed_audio(r[1])
                mask[b, :lx + ly, :lx + ly] = o.single_mask(lx, ly)
            o.prefill(xyo, mask, B, 0)
            if got is None:
                rt = m._rt[B]
                with torch.inference_mode():
                    for t, oc in ((rt["k"], o.cache[B][0]), (rt["v"], o.cache[B][1])):
                        t.copy_(torch.from_numpy(oc).to(dev).to(t.dtype))
                got = m.decode_hidden(B, _T(xin, dev)).cpu().numpy()
            dist[(B, name)] = float(np.abs(got - o.decode(xin, B, kv)).mean())
        assert dist[(B, "same")] < 2e-4, dist
        del m
    print("mean |hidden - bf16 oracle| with the library's slice count / with the other one:", dist)
    assert any(dist[(B, "other")] > 1.2 * dist[(B, "same")] for B in (4, 5)), dist
