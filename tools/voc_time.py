"""flow + Generator pass time: one utterance (T=500, broadcast ge) and a time-concatenated batch of 10 (per-frame ge), per model version"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import _VocoderNative
dev = torch.device("cuda:0")
for ver in (sys.argv[1:] or ["v2Pro", "v2ProPlus"]):
    hps = synth.sovits_hps(ver)
    sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
    voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.float32 if os.environ.get("GSV_VOC_DTYPE") == "fp32" else torch.bfloat16, dev)
    for T, per_frame in ((500, False), (5800, True)):
        ge = torch.from_numpy(synth.synth_ge(0, hps["model"]["gin_channels"])).to(dev)
        if per_frame:
            ge = ge.expand(-1, -1, T).contiguous()
        z = torch.randn(1, 192, T, device=dev); m = torch.ones(1, 1, T, device=dev)
        for _ in range(3): voc.flow_dec(z, m, ge)
        n = 20 if T == 500 else 5
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(n): voc.flow_dec(z, m, ge)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n)
        print("%s T=%d: %.3f ms per pass = %.3f ms per 10 s of audio" % (ver, T, best * 1e3, best * 1e3 * 500 / T), flush=True)
