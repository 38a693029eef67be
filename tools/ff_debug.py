"""print flowfuse phase stamps (GSV_FF_DEBUG=1): cycles between stamps for block 0 / wave 0"""
import os, sys
os.environ["GSV_FF_DEBUG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gsv-tts-lite_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import _VocoderNative
dev = torch.device("cuda:0")
hps = synth.sovits_hps("v2Pro")
w = synth.sovits_weights(hps, seed=1, hot_path_only=True)
v = _VocoderNative(hps["model"], {k: torch.from_numpy(a) for k, a in w.items()}, torch.bfloat16, dev)
T = 500
z = torch.from_numpy(synth.hashed_uniform("z", (1, 192, T), 1)).to(dev)
ge = torch.from_numpy(synth.synth_ge(0, 1024, 1)).to(dev)
for i in range(3):
    print("--- call", i, file=sys.stderr)
    v.flow(z, torch.ones(1, 1, T, device=dev), ge)
