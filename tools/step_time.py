"""raw decode step time at batch B (graph replay): python tools/step_time.py B [bf16|fp8|fp32]; GSV_BATCHED_MIN picks the path"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gsv-tts-lite_amd"))
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
B = int(sys.argv[1]); dev = torch.device("cuda:0")
dt = {"bf16": torch.bfloat16, "fp8": torch.float8_e4m3fn, "fp32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
cfg = synth.gpt_config(n_layer=int(os.environ.get('GSV_NLAYER', '24')))
m = Text2SemanticDecoder(cfg); m.load_state_dict(synth.gpt_weights(cfg, seed=1, eos_gain=-8.0))
m.initialize_runtime(dt, dev, [(B, int(os.environ.get('GSV_T', '512')))])
if os.environ.get('GSV_NO_GRAPH'):      # eager launches (a counter pass of the graph-replayed chain hung under rocprofv3 --pmc FETCH_SIZE)
    m.use_graph = False
NST = int(os.environ.get('GSV_STEPS', '100'))
rs = [synth.synth_request(i, 40, 60, int(os.environ.get('GSV_PROMPT_TOK', '100')), seed=1) for i in range(B)]   # kv ~ 100 + this + the steps run
with torch.inference_mode():
    xy, xl, yl, _, _ = m.embed_prompt([torch.from_numpy(r[0]).to(dev) for r in rs], [torch.from_numpy(r[1]).to(dev) for r in rs], [torch.from_numpy(r[2]).to(dev) for r in rs])
    m.prefill(B, 0, xy, xl, yl)
    m._set_ctl(m._rt[B], 0, 0, False, 1.0)
    m._decode(B, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); m._decode(B, NST); torch.cuda.synchronize()
    print("B=%d %s step %.3f ms (%s, batched_min %d)" % (B, sys.argv[2] if len(sys.argv) > 2 else "bf16", (time.perf_counter() - t0) * 1e3 / NST,
                                                     "batched chain" if B >= m.batched_min else "per-sequence kernels", m.batched_min))
