"""shader-clock stamps of block (0, 0) of the last layer's batched attention kernel: python tools/battn_stamps.py B [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsv-tts-lite_amd"))
import numpy as np, torch
from gsv_tts_lite_amd import synth, _native as N
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
B = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
cfg = synth.gpt_config(); m = Text2SemanticDecoder(cfg); m.load_state_dict(synth.gpt_weights(cfg, seed=1, eos_gain=-8.0))
m.initialize_runtime(torch.bfloat16, dev, [(B, T)])
rs = [synth.synth_request(i, 40, 60, 100, seed=1) for i in range(B)]
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
N.check(N.lib().gsv_t2s_set_debug(m._h, dbg.data_ptr()))
with torch.inference_mode():
    xy, xl, yl, _, _ = m.embed_prompt([torch.from_numpy(r[0]).to(dev) for r in rs], [torch.from_numpy(r[1]).to(dev) for r in rs], [torch.from_numpy(r[2]).to(dev) for r in rs])
    m.prefill(B, 0, xy, xl, yl)
    m._set_ctl(m._rt[B], 0, 0, False, 1.0)
    m._decode(B, 20); torch.cuda.synchronize()
    rows = []
    for it in range(20):
        m._decode(B, 1); torch.cuda.synchronize()
        rows.append(dbg.cpu().numpy().astype(np.int64).copy())
r = np.array(rows)
print("B=%d T=%d attention block (0,0), cycles since entry: kv_len landed %d | q row landed %d | barrier 1 passed %d | scores + max %d | P.V + reductions %d | end %d" %
      ((B, T) + tuple(int(np.median(r[:, i] - r[:, 0])) for i in range(1, 7))))
