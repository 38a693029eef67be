import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gsv-tts-lite_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, torch.nn.functional as F
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.sovits import SynthesizerTrn
dev = torch.device("cuda:0")
hps = synth.sovits_hps("v2Pro")
vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"]); vq.load_state_dict(synth.sovits_weights(hps, seed=7))
vq.initialize_runtime(torch.bfloat16, dev, [64])
from oracle.sovits_encoder import TextEncoder, codebook_decode
enc = TextEncoder(vq.hps_model, vq._weights, dev)
rng = np.random.default_rng(3)
for n, P in [(25, 30), (150, 100), (250, 100)]:
    codes = torch.from_numpy(rng.integers(0, 1024, (1, 1, n))).to(dev); text = torch.from_numpy(rng.integers(1, 700, (1, P))).to(dev)
    ge = torch.from_numpy(synth.synth_ge(1, 1024, 7)).to(dev); ge_in = enc.ge_to512(ge)
    with torch.inference_mode():
        q = F.interpolate(codebook_decode(vq._weights, codes), size=2 * n, mode="nearest")
        m_ref, l_ref, _ = enc.infer(q, text, ge_in, 1); a_ref = enc.mrte.cross_attention.attn[0].clone()
        m, l, a = vq._voc.enc_p(codes[0, 0], text[0], ge_in)
    print(n, P, "m_p max %.4f mean %.5f (|ref| mean %.3f)  logs max %.4f mean %.5f  attn max %.4f" % (
        (m - m_ref).abs().max(), (m - m_ref).abs().mean(), m_ref.abs().mean(), (l - l_ref).abs().max(), (l - l_ref).abs().mean(), (a - a_ref).abs().max()))
