"""Time gsv_align_viterbi (device) for a 10 s utterance and a time-concatenated batch; for scale, the same
path as a per-frame loop of tensor ops (what the reference's TTS._viterbi_monotonic issues) is ~6 launches/frame."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch
from gsv_tts_lite_amd import subtitles as sub, synth
dev = torch.device("cuda:0")
for T, P in [(500, 100), (500, 250), (5000, 1000)]:
    a = torch.from_numpy(synth.synth_attn(1, 4, T, P, 3, 10)).to(dev)
    for _ in range(3): sub.viterbi_monotonic(a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): sub.viterbi_monotonic(a)
    torch.cuda.synchronize()
    print("T=%d N=%d: %.1f us per alignment" % (T, P, (time.perf_counter() - t) / 20 * 1e6))
