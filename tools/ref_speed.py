"""Time and error of the reference-audio path on device (gsv_ref_*): spectrogram, get_ge, extract_latent."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import SynthesizerTrn
from oracle import oracle as orc
dev = torch.device("cuda:0")
hps = synth.sovits_hps("v2Pro")
vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
sd = dict(synth.sovits_weights(hps, hot_path_only=True)); sd.update(synth.ref_audio_weights(hps))
vq.load_state_dict(sd); vq.initialize_runtime(torch.bfloat16, dev, [50])
o = orc.RefAudioOracle(synth.ref_audio_weights(hps))
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for secs in (3, 10):
    a = synth.synth_audio(0, 32000 * secs); at = torch.from_numpy(a).to(dev)
    sv = synth.synth_sv_emb(0); svt = torch.from_numpy(sv).to(dev)
    ssl = synth.synth_ssl(0, 50 * secs); st = torch.from_numpy(ssl).to(dev)
    spec = vq.spectrogram(at); so = orc.spectrogram(a)
    ge = vq.get_ge(spec, svt); go = o.get_ge(so, sv)
    codes, mg = vq._ref_audio().extract_latent(st, return_margin=True); co, mo = o.extract_latent(ssl[0])
    t0 = time.perf_counter(); orc.spectrogram(a); o.get_ge(so, sv); o.extract_latent(ssl[0]); tc = (time.perf_counter() - t0) * 1e3
    print("%2d s ref audio: spectrogram %.3f ms (err %.1e of peak %.0f), get_ge %.3f ms (err %.1e), extract_latent %.3f ms "
          "(codes equal %d/%d, min margin %.3f); numpy oracle %.0f ms" % (
              secs, timeit(lambda: vq.spectrogram(at)), np.abs(spec[0].cpu().numpy() - so).max() / so.max(), so.max(),
              timeit(lambda: vq.get_ge(spec, svt)), np.abs(ge[0, :, 0].cpu().numpy() - go).max(),
              timeit(lambda: vq.extract_latent(st)), int((codes[0, 0].cpu().numpy() == co).sum()), len(co), mo.min(), tc))
