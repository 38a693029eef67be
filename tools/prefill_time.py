"""packed prompt pass of B requests (bench cb lengths): host wall time of embed / prefill / first step; run under rocprofv3 for the kernel table"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gsv-tts-lite_amd"))
import torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = synth.gpt_config()
m = Text2SemanticDecoder(cfg); m.load_state_dict(synth.gpt_weights(cfg, seed=1)); m.initialize_runtime(torch.float32 if os.environ.get("GSV_DTYPE") == "fp32" else torch.bfloat16, dev, [(B, 512), (B, 1024)])
lens = synth.mixed_lengths(B) if B > 1 else [(60, 100)]      # one prompt: the bench shape (40 + 60 phonemes, 100 prompt tokens)
rs = [synth.synth_request(i, 40, t, n) for i, (t, n) in enumerate(lens)]
X = [torch.from_numpy(r[0]).to(dev) for r in rs]; Y = [torch.from_numpy(r[1]).to(dev) for r in rs]; Bt = [torch.from_numpy(r[2]).to(dev) for r in rs]
with torch.inference_mode():
    m._set_ctl(m._rt[B], 0, 0, False, 1.0)
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        xy, xl, yl, _, _ = m.embed_prompt(X, Y, Bt)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        m.prefill(B, 0, xy, xl, yl)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        m._decode(B, 1)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print("B=%d rows %d x l_max %d = %d: embed %.2f ms, prompt pass %.2f ms, first step %.2f ms" % (B, B, xy.shape[1], B * xy.shape[1], (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
