"""one long flow+Generator pass (a time-concatenated batch of 10 utterances, ~116 s of audio) for profiling"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import _VocoderNative
ver = sys.argv[1] if len(sys.argv) > 1 else "v2ProPlus"; T = int(sys.argv[2]) if len(sys.argv) > 2 else 5800
dev = torch.device("cuda:0"); hps = synth.sovits_hps(ver)
sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.bfloat16, dev)
ge = torch.from_numpy(synth.synth_ge(0, hps["model"]["gin_channels"])).to(dev).expand(-1, -1, T).contiguous()
z = torch.randn(1, 192, T, device=dev); m = torch.ones(1, 1, T, device=dev)
for _ in range(2): voc.flow_dec(z, m, ge)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): voc.flow_dec(z, m, ge)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print("%s T=%d: %.2f ms per pass = %.2f ms per 10 s" % (ver, T, dt * 1e3, dt * 1e3 * 500 / T))
