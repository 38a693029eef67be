#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs 2/3 (not the driver's bench.py contract):
continuous batching at bs=B over mixed-length synthetic utterances on one MI355X.
    python bench_configs.py --batch 32 --requests 256 --dtype bf16
Prints one JSON line: aggregate semantic tokens/s of t2s.infer_batched (prefills + refills included)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--requests", type=int, default=256)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--max-kv", type=int, default=512)
ap.add_argument("--eos-gain", type=float, default=4.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
cfg = synth.gpt_config()
m = Text2SemanticDecoder(cfg)
m.load_state_dict(synth.gpt_weights(cfg, seed=1234, eos_gain=a.eos_gain))
m.initialize_runtime(dtype, dev, [(a.batch, a.max_kv // 2), (a.batch, a.max_kv)])
lens = synth.mixed_lengths(a.requests)
reqs = [synth.synth_request(i, 40, t, n) for i, (t, n) in enumerate(lens)]
xs = [torch.from_numpy(r[0]).to(dev) for r in reqs]
ys = [torch.from_numpy(r[1]).to(dev) for r in reqs]
bs = [torch.from_numpy(r[2]).to(dev) for r in reqs]
m.infer_batched(xs[: a.batch], ys[: a.batch], bs[: a.batch], top_k=1)   # warm-up (graph capture)
torch.cuda.synchronize()
t0 = time.perf_counter()
pred, orig = m.infer_batched(xs, ys, bs, top_k=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ntok = int(sum(len(p) for p in pred))
print(json.dumps({"config": "continuous batching bs=%d, %d mixed-length requests, %s" % (a.batch, a.requests, a.dtype),
                  "tokens": ntok, "seconds": dt, "tokens_per_s": ntok / dt, "requests_per_s": a.requests / dt,
                  "mean_tokens_per_request": ntok / a.requests}))
