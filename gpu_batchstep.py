"""scratch: raw decode-step time at several batch sizes (graph replay, no host sync inside)"""
import os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsv-tts-lite_amd"))
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
dev = torch.device("cuda:0")
cfg = synth.gpt_config(); w = synth.gpt_weights(cfg, eos_gain=0.0)
def run(B):
    m = Text2SemanticDecoder(cfg); m.load_state_dict(w); m.initialize_runtime(torch.bfloat16, dev, [(B, 512)])
    reqs = [synth.synth_request(i) for i in range(B)]
    T = lambda a: torch.from_numpy(a).to(dev)
    xy, xl, yl, _, _ = m.embed_prompt([T(r[0]) for r in reqs], [T(r[1]) for r in reqs], [T(r[2]) for r in reqs])
    rt = m._rt[B]; m._set_ctl(rt, False, 0, False, 1.0)
    m.prefill(B, 0, xy, xl, yl)
    m._decode(B, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); m._decode(B, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("B=%d  %.3f ms/step  %.0f tok/s" % (B, dt * 10, B * 100 / dt))
    del m

with torch.inference_mode():
    for B in (1, 4, 8, 16, 32, 64):
        run(B)
