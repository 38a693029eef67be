"""stress: how often does the persistent step disagree with itself / with the per-layer path"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsv-tts-lite_amd"))
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cfg = synth.gpt_config(n_layer=24)
w = synth.gpt_weights(cfg, seed=1234, eos_gain=1.0)
m = Text2SemanticDecoder(cfg); m.load_state_dict(w); m.initialize_runtime(torch.float32, dev, [(1, 128), (1, 160), (4, 160)])
x, y, b, _ = synth.synth_request(0, 12, 24, 30)
m.use_megastep = False
ref = m.infer(T(x)[None], T(y)[None], T(b)[None], top_k=1)[0, 0].cpu().numpy()
m.use_megastep = True
bad = 0; N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t0 = time.time()
for i in range(N):
    m.use_graph = (i % 2 == 0)
    tok = m.infer(T(x)[None], T(y)[None], T(b)[None], top_k=1)[0, 0].cpu().numpy()
    if not np.array_equal(tok, ref):
        bad += 1
        nm = min(len(tok), len(ref)); print("run", i, "graph", m.use_graph, "first diff", int(np.argmax(tok[:nm] != ref[:nm])), flush=True)
print("mismatching runs: %d / %d  (%d steps each)  err=%s  %.1fs" % (bad, N, len(ref), m.megastep_error(), time.time() - t0))
