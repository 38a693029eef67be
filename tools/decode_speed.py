"""time of SynthesizerTrn.decode (quantizer + torch enc_p + HIP flow/Generator) vs its HIP part alone, bf16, 10 s"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gsv-tts-lite_amd"))
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import SynthesizerTrn
dev = torch.device("cuda:0")
hps = synth.sovits_hps("v2Pro")
vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
vq.load_state_dict(synth.sovits_weights(hps, seed=1))
vq.initialize_runtime(torch.bfloat16, dev, [500])
codes = torch.randint(0, 1024, (1, 1, 250), device=dev)
text = torch.randint(1, 700, (1, 100), device=dev)
ge = torch.from_numpy(synth.synth_ge(0, 1024, 1)).to(dev)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
full = t(lambda: vq.decode(codes, text, ge, noise_scale=0.5))
z = torch.randn(1, 192, 500, device=dev); m = torch.ones(1, 1, 500, device=dev)
hip = t(lambda: vq.flow_dec(z, m, ge))
print("decode() %.2f ms   flow_dec (HIP) %.2f ms   -> quantizer + enc_p + glue (torch) %.2f ms" % (full, hip, full - hip))
