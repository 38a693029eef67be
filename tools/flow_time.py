"""the flow alone (gsv_voc_flow: 4 coupling layers in reverse) at several frame counts; GSV_FLOW_STAGED_MAX_T=0 forces the fused kernel,
GSV_FLOW_STAGED_RPB the frame tiles per block of the staged form, GSV_FLOW_MERGED_MAX_T=0 the ten-launch form instead of the merged one.  usage: tools/flow_time.py [T ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import _VocoderNative
dev = torch.device("cuda:0")
hps = synth.sovits_hps("v2Pro")
sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.bfloat16, dev)
for T in [int(t) for t in sys.argv[1:]] or [50, 500, 1000, 2000, 5800]:
    for per_frame in (False, True):
        ge = torch.from_numpy(synth.synth_ge(0, 1024)).to(dev)
        if per_frame:
            ge = ge.expand(-1, -1, T).contiguous()
        z = torch.randn(1, 192, T, device=dev); m = torch.ones(1, 1, T, device=dev)
        for _ in range(3): voc.flow(z, m, ge)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): voc.flow(z, m, ge)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 20)
        print("flow T=%d %s ge: %.1f us (staged_max_T=%s rpb=%s)" % (T, "per-frame" if per_frame else "broadcast", best * 1e6,
              os.environ.get("GSV_FLOW_STAGED_MAX_T", "default"), os.environ.get("GSV_FLOW_STAGED_RPB", "auto")), flush=True)
