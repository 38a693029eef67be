"""decode-step time of N freshly built model instances in one process (bs = 1, bf16, the bench's buckets): the spread between
instances is the placement lottery DESIGN.md discusses.  python tools/placement_ab.py [N] ; knobs: GSV_NO_ARENA, GSV_ARENA_ALIGN,
GSV_STATE_SEPARATE=1 (state tensors as separate torch allocations instead of one 64 KB-aligned block)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gsv-tts-lite_amd"))
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = synth.gpt_config()
w = synth.gpt_weights(cfg, seed=1234, eos_gain=0.0)
ts, keep = [], []
for i in range(N):
    m = Text2SemanticDecoder(cfg); m.load_state_dict(w)
    m.initialize_runtime(torch.bfloat16, dev, [(1, 512), (1, 1024)], tune_placement=1)
    with torch.inference_mode():
        t = m._time_step(1)
    ts.append(t)
    if os.environ.get("KEEP"): keep.append(m)       # keep the instances alive: the next one cannot reuse the same addresses
    else: del m
print("step ms per instance:", " ".join("%.4f" % t for t in ts), "| min %.4f max %.4f mean %.4f" % (min(ts), max(ts), sum(ts) / len(ts)))
