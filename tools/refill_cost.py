"""What one refill's prompt pass costs the slot loop's steps: the step time in steady state, the steps right after a prompt pass
that ran alone (cache residency), and steps with a prompt pass beside them on another stream (sharing the chip).

    python tools/refill_cost.py [B=32]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gsv-tts-lite_amd"))
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = synth.gpt_config()
m = Text2SemanticDecoder(cfg); m.load_state_dict(synth.gpt_weights(cfg, seed=1, eos_gain=-8.0)); m.initialize_runtime(torch.bfloat16, dev, [(B, 1024)])
lens = synth.mixed_lengths(B)
rs = [synth.synth_request(i, 40, t, n) for i, (t, n) in enumerate(lens)]
X = [torch.from_numpy(r[0]).to(dev) for r in rs]; Y = [torch.from_numpy(r[1]).to(dev) for r in rs]; Bt = [torch.from_numpy(r[2]).to(dev) for r in rs]
ev = lambda: torch.cuda.Event(enable_timing=True)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
with torch.inference_mode():
    m._set_ctl(m._rt[B], 0, 0, False, 1.0)
    xy, xl, yl, _, _ = m.embed_prompt(X, Y, Bt)
    m.prefill(B, 0, xy, xl, yl)
    m._decode(B, 20); torch.cuda.synchronize()
    # 1. steady state
    a, b = ev(), ev(); a.record(); m._decode(B, 50); b.record(); torch.cuda.synchronize()
    steady = a.elapsed_time(b) / 50
    print("B=%d steady step %.4f ms" % (B, steady))
    # the prompt pass of k requests, embedded once
    for k in (1, 2, 4):
        xyk, xlk, ylk, _, _ = m.embed_prompt(X[:k], Y[:k], Bt[:k])
        slk = torch.tensor(list(range(k)), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        # 2. pass alone (serial, on the steps' stream), then five single steps
        tp, after = [], [[] for _ in range(6)]
        for it in range(12):
            e = [ev() for _ in range(8)]
            e[0].record(); m.prefill_slots(B, list(range(k)), xyk, xlk, ylk); e[1].record()
            for j in range(6):
                m._decode(B, 1); e[2 + j].record()
            torch.cuda.synchronize()
            if it >= 2:
                tp.append(e[0].elapsed_time(e[1]))
                for j in range(6): after[j].append(e[1 + j].elapsed_time(e[2 + j]))
        mean = lambda v: sum(v) / len(v)
        print("k=%d prompt pass alone %.3f ms (%d rows); steps after it: %s ms (steady %.4f): a pass costs the steps %.3f ms beyond its own time" % (
            k, mean(tp), k * xyk.shape[1], " ".join("%.4f" % mean(x) for x in after), steady, sum(mean(x) - steady for x in after)))
        # 3. pass on the side stream beside a window of 5 steps
        tw = []
        for it in range(12):
            a, b = ev(), ev()
            torch.cuda.synchronize()
            a.record(main)
            side.wait_stream(main)
            m.prefill_slots_staged(B, slk, xyk, xlk, ylk, side.cuda_stream)
            m._decode(B, 10)
            main.wait_stream(side)
            b.record(main); torch.cuda.synchronize()
            if it >= 2: tw.append(a.elapsed_time(b))
        print("k=%d pass on a side stream beside 10 steps: %.3f ms (10 steady steps %.3f, + the pass alone %.3f = %.3f)" % (
            k, mean(tw), 10 * steady, mean(tp), 10 * steady + mean(tp)))
