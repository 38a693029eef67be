"""Where the driver-visible vocoder time goes beyond the kernels: one flow + Generator pass started on an IDLE stream (what bench.py's
vocoder leg is) against the same pass queued back to back; host time in front of / inside the C call."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch
from gsv_tts_lite_amd import synth, _native as N
from gsv_tts_lite_amd.sovits import _VocoderNative
dev = torch.device("cuda:0")
for ver in (sys.argv[1:] or ["v2Pro", "v2ProPlus"]):
    hps = synth.sovits_hps(ver)
    sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
    voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.bfloat16, dev)
    T = 500
    ge = torch.from_numpy(synth.synth_ge(0, hps["model"]["gin_channels"])).to(dev)
    z = torch.randn(1, 192, T, device=dev); m = torch.ones(1, 1, T, device=dev)
    for _ in range(5): voc.flow_dec(z, m, ge)
    torch.cuda.synchronize()
    # (a) idle start, events around the call (bench.py's vocoder leg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev, host, wall = [], [], []
    for _ in range(20):
        torch.cuda.synchronize(); time.sleep(0.002)
        t0 = time.perf_counter(); e0.record(); voc.flow_dec(z, m, ge); t1 = time.perf_counter(); e1.record()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ev.append(e0.elapsed_time(e1)); host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
    ev.sort(); host.sort(); wall.sort()
    # (b) back to back
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): voc.flow_dec(z, m, ge)
    torch.cuda.synchronize(); b2b = (time.perf_counter() - t) / 20 * 1e3
    # (c) the C call alone (arguments prepared)
    zz, gg, T_, Tg = voc._prep(z, ge); mk = m.reshape(-1).contiguous(); out = torch.empty(1, 1, T * voc.samples_per_frame, device=dev); ws = voc._workspace(T)
    sp = N.current_stream_ptr(dev); L = N.lib()
    cc = []
    for _ in range(20):
        torch.cuda.synchronize(); time.sleep(0.002)
        t0 = time.perf_counter()
        N.check(L.gsv_voc_flow_dec(voc._h, zz.data_ptr(), mk.data_ptr(), gg.data_ptr(), T, Tg, out.data_ptr(), ws.data_ptr(), ws.numel(), sp))
        cc.append((time.perf_counter() - t0) * 1e3)
    cc.sort()
    print("%s T=%d: idle start: events %.3f ms (min %.3f), host time in flow_dec %.3f ms, wall incl. sync %.3f | back to back %.3f ms | the C call alone (enqueue) %.3f ms"
          % (ver, T, ev[len(ev) // 2], ev[0], host[len(host) // 2], wall[len(wall) // 2], b2b, cc[len(cc) // 2]), flush=True)
