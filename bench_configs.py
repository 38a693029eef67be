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
ap.add_argument("--vocoder", default="", help="v2 | v2Pro | v2ProPlus: also run flow + Generator over every utterance, "
                "time-concatenated in batches of --sovits-batch like TTS.infer_batched (TTS.py:728-764)")
ap.add_argument("--sovits-batch", type=int, default=10)
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
out = {"config": "continuous batching bs=%d, %d mixed-length requests, %s" % (a.batch, a.requests, a.dtype),
       "tokens": ntok, "seconds": dt, "tokens_per_s": ntok / dt, "requests_per_s": a.requests / dt,
       "mean_tokens_per_request": ntok / a.requests}
if a.vocoder:
    # flow + Generator on z ~ N(0,1) of the generated lengths (enc_p's output statistics are not what is timed here),
    # per-frame speaker embedding of the time-concatenated batch, as infer_batched feeds it
    from gsv_tts_lite_amd.sovits import _VocoderNative
    hps = synth.sovits_hps(a.vocoder)
    sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
    voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, dtype, dev)
    gin = hps["model"]["gin_channels"]
    ge1 = torch.from_numpy(synth.synth_ge(0, gin)).to(dev)
    frames = [2 * len(p) for p in pred]
    order = sorted(range(len(frames)), key=lambda i: frames[i])
    batches = [order[i:i + a.sovits_batch] for i in range(0, len(order), a.sovits_batch)]
    def run():
        tot = 0
        for b in batches:
            T = sum(frames[i] for i in b)
            z = torch.randn(1, 192, T, device=dev)
            voc.flow_dec(z, torch.ones(1, 1, T, device=dev), ge1.expand(-1, -1, T).contiguous())
            tot += T
        return tot
    run(); torch.cuda.synchronize()
    t1 = time.perf_counter(); tot = run(); torch.cuda.synchronize(); dv = time.perf_counter() - t1
    out.update({"vocoder": a.vocoder, "vocoder_seconds": dv, "audio_s": tot / 50.0, "vocoder_audio_s_per_s": tot / 50.0 / dv,
                "end_to_end_tokens_per_s": ntok / (dt + dv), "end_to_end_audio_s_per_s": tot / 50.0 / (dt + dv)})
print(json.dumps(out))
