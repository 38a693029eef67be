"""ms/token of the greedy (device loop) vs the default stochastic (top_k=15, rep 1.35) AR path at bs=1"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gsv-tts-lite_amd"))
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
dev = torch.device("cuda:0")
cfg = synth.gpt_config()
m = Text2SemanticDecoder(cfg)
m.load_state_dict(synth.gpt_weights(cfg, seed=1234, eos_gain=0.0))
m.initialize_runtime(torch.bfloat16, dev, [(1, 256), (1, 450)])
x, y, bert, _ = synth.synth_request(0, 40, 60, 100, seed=1234)
x, y, bert = (torch.from_numpy(t)[None].to(dev) for t in (x, y, bert))
for name, kw in [("greedy", dict(top_k=1)), ("top_k=15 rep=1.35", dict(top_k=15, repetition_penalty=1.35)),
                 ("top_k=15 top_p=0.9", dict(top_k=15, top_p=0.9))]:
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tok = m.infer(x, y, bert, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = tok.shape[-1]
    print("%-22s %4d tokens  %.3f ms/token (incl. prefill)" % (name, n, dt * 1e3 / max(n, 1)))
