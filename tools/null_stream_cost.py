"""Eager launches on the null stream against a stream of the pool: the prompt pass of one request (TTFT) and one vocoder pass.
    python tools/null_stream_cost.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
from gsv_tts_lite_amd.sovits import _VocoderNative
dev = torch.device("cuda:0")
cfg = synth.gpt_config()
m = Text2SemanticDecoder(cfg); m.load_state_dict(synth.gpt_weights(cfg, seed=1, eos_gain=-8.0)); m.initialize_runtime(torch.bfloat16, dev, [(1, 512), (1, 1024)])
r = synth.synth_request(0, 100, 60, 100, seed=1)
X, Y, Bt = [torch.from_numpy(r[0]).to(dev)], [torch.from_numpy(r[1]).to(dev)], [torch.from_numpy(r[2]).to(dev)]
hps = synth.sovits_hps("v2Pro")
sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.bfloat16, dev)
ge = torch.from_numpy(synth.synth_ge(0, hps["model"]["gin_channels"])).to(dev)
z = torch.randn(1, 192, 500, device=dev); mk = torch.ones(1, 1, 500, device=dev)
def med(f, n=30):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return sorted(ts[3:])[len(ts[3:]) // 2] * 1e3
def ttft():
    xy, xl, yl, _, _ = m.embed_prompt(X, Y, Bt); m.prefill(1, 0, xy, xl, yl); m._decode(1, 1)
with torch.inference_mode():
    m._set_ctl(m._rt[1], 0, 0, False, 1.0)
    for name, st in (("null stream", torch.cuda.default_stream(dev)), ("pool stream", torch.cuda.Stream(device=dev)), ("high-priority pool stream", torch.cuda.Stream(device=dev, priority=-1))):
        with torch.cuda.stream(st):
            print("%-26s prompt pass + first step %.3f ms | vocoder pass %.3f ms | one pass of 10 back to back %.3f ms" % (
                name, med(ttft), med(lambda: voc.flow_dec(z, mk, ge)), med(lambda: [voc.flow_dec(z, mk, ge) for _ in range(10)], 8) / 10), flush=True)
