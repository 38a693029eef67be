"""one SynthesizerTrn.decode call on a time-concatenated batch (10 utterances, per-token ge, slice_indices), as TTS.infer_batched issues it"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch, numpy as np
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import SynthesizerTrn
ver = sys.argv[1] if len(sys.argv) > 1 else "v2ProPlus"
dev = torch.device("cuda:0"); hps = synth.sovits_hps(ver)
vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
vq.load_state_dict(synth.sovits_weights(hps, seed=1234)); vq.initialize_runtime(torch.bfloat16, dev, [])
rng = np.random.default_rng(1)
lens = rng.integers(50, 400, 10); ph = rng.integers(20, 120, 10)
n, P = int(lens.sum()), int(ph.sum())
codes = torch.from_numpy(rng.integers(0, 1024, (1, 1, n))).to(dev); text = torch.from_numpy(rng.integers(1, 700, (1, P))).to(dev)
ge = torch.from_numpy(synth.synth_ge(0, hps["model"]["gin_channels"])).to(dev).expand(-1, -1, n).contiguous()
ends = np.cumsum(ph); pairs = np.stack([ends - ph, ends], 1)
sl = torch.from_numpy(np.repeat(pairs, 2 * lens, axis=0).astype(np.int64)).to(dev)
for _ in range(2): vq.decode(codes, text, ge, noise_scale=0.5, cuda_graph=False, slice_indices=sl)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): vq.decode(codes, text, ge, noise_scale=0.5, cuda_graph=False, slice_indices=sl)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print("%s decode of %d tokens (%d frames, %d phonemes): %.2f ms = %.2f ms per 10 s" % (ver, n, 2 * n, P, dt * 1e3, dt * 1e3 * 250 / n))
