import os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsv-tts-lite_amd"))
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
dev = torch.device("cuda:0")
cfg = synth.gpt_config(); w = synth.gpt_weights(cfg, eos_gain=0.0)
m = Text2SemanticDecoder(cfg); m.load_state_dict(w); m.initialize_runtime(torch.bfloat16, dev, [(1, 256), (1, 450)])
x, y, b, _ = synth.synth_request(0); T = lambda a: torch.from_numpy(a).to(dev)
for tk in (1, 15):
    m.infer(T(x)[None], T(y)[None], T(b)[None], top_k=tk); torch.cuda.synchronize()
    t0 = time.perf_counter(); tok = m.infer(T(x)[None], T(y)[None], T(b)[None], top_k=tk); torch.cuda.synchronize()
    print("top_k=%d: %d tokens %.1f ms -> %.0f tok/s" % (tk, tok.shape[-1], (time.perf_counter()-t0)*1e3, tok.shape[-1]/(time.perf_counter()-t0)))
