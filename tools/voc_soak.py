"""determinism soak of flow + Generator: the same input must give bit-identical output on every pass (a missing
barrier or a cross-wave LDS race in wconv / wups / flowfuse would show up as a flicker)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import _VocoderNative
dev = torch.device("cuda:0")
bad = 0
for ver in ("v2Pro", "v2ProPlus", "v2"):
    hps = synth.sovits_hps(ver)
    sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
    voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.bfloat16, dev)
    for T, n in ((500, 150), (131, 150), (1777, 40)):
        g = torch.Generator(device=dev); g.manual_seed(T)
        z = torch.randn(1, 192, T, device=dev, generator=g); m = torch.ones(1, 1, T, device=dev)
        ge = torch.from_numpy(synth.synth_ge(0, hps["model"]["gin_channels"])).to(dev)
        ref = voc.flow_dec(z, m, ge).clone()
        diff = 0
        for _ in range(n):
            diff += int((voc.flow_dec(z, m, ge) != ref).sum())
        print(ver, "T=%d" % T, "passes", n, "differing samples", diff, "finite", bool(torch.isfinite(ref).all()))
        bad += diff
print("SOAK", "OK" if bad == 0 else "FAILED")
