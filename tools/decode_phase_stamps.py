"""scratch: per-phase shader-clock breakdown of the last layer's decode kernels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsv-tts-lite_amd"))
import numpy as np, torch
from gsv_tts_lite_amd import synth, _native as N
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
dev = torch.device("cuda:0")
cfg = synth.gpt_config(); w = synth.gpt_weights(cfg, eos_gain=0.0)
m = Text2SemanticDecoder(cfg); m.load_state_dict(w); m.initialize_runtime(torch.bfloat16, dev, [(1, 256), (1, 450)])
x, y, bert, _ = synth.synth_request(0)
T = lambda a: torch.from_numpy(a).to(dev)
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
N.check(N.lib().gsv_t2s_set_debug(m._h, dbg.data_ptr()))
tok = m.infer(T(x)[None], T(y)[None], T(bert)[None], top_k=1)
torch.cuda.synchronize()
rows = []
for it in range(20):
    m._decode(1, 1); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.int64)
    rows.append(d.copy())
r = np.array(rows)
a = r[:, 1:7] - r[:, 0:6]
f = r[:, 9:13] - r[:, 8:12]
print("attn phases (cycles) entry->firstload, LN+xs, QKV, attention, combine, panel:", np.median(a, axis=0), "total", np.median(r[:, 6] - r[:, 0]))
print("ffn phases entry->firstload, LN+xs, W1, panel:", np.median(f, axis=0), "total", np.median(r[:, 12] - r[:, 8]))
print("attn end -> ffn start:", np.median(r[:, 8] - r[:, 6]))
print("ffn LN detail: park", np.median(r[:,13]-r[:,9]), "barrier", np.median(r[:,14]-r[:,13]), "finish+ln", np.median(r[:,15]-r[:,14]), "xs+barrier", np.median(r[:,10]-r[:,15]))
w15 = r[:, 16:]
print("last wave, relative to wave 0's entry stamp: attn stamps 0..6:", np.median(w15[:, 0:7] - r[:, 0:1], axis=0))
print("wave 0,    relative to its entry stamp:      attn stamps 0..6:", np.median(r[:, 0:7] - r[:, 0:1], axis=0))
print("last wave ffn stamps 8,9,13,14,15,10,11,12 rel. to wave 0's stamp 8:", np.median(w15[:, [8, 9, 13, 14, 15, 10, 11, 12]] - r[:, 8:9], axis=0))
print("wave 0    ffn stamps 8,9,13,14,15,10,11,12 rel. to its stamp 8:     ", np.median(r[:, [8, 9, 13, 14, 15, 10, 11, 12]] - r[:, 8:9], axis=0))
