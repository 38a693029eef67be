"""scratch GPU bring-up script (not a test): prints HIP-vs-oracle diffs stage by stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsv-tts-lite_amd"))
import numpy as np, torch
from gsv_tts_lite_amd import synth
from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
from gsv_tts_lite_amd.sovits import _VocoderNative
from oracle import oracle as orc

dev = torch.device("cuda:0")
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

def t2s_layers(dtype):
    g = np.load(os.path.join(ROOT, "tests/golden/t2s_layers.npz"))
    cfg = synth.gpt_config(n_layer=3); w = synth.gpt_weights(cfg, seed=int(g["seed"]))
    o = orc.T2SOracle(cfg, w, [(1, 96), (2, 96)])
    m = Text2SemanticDecoder(cfg); m.load_state_dict(w); m.initialize_runtime(dtype, dev, [(1, 96), (2, 96)])
    x, y, bert = g["s_x"], g["s_y"], g["s_bert"]
    xy, xl, yl, _, _ = m.embed_prompt([T(x)], [T(y)], [T(bert)])
    print(dtype, "embed diff", abs(xy.cpu().numpy() - g["s_xy"]).max())
    L = len(x) + len(y)
    m.prefill(1, 0, xy, xl, yl)
    torch.cuda.synchronize()
    hid = m._rt[1]["hidden"].cpu().numpy()
    print(" prefill last-hidden diff", abs(hid[0] - g["s_hidden"][0, -1]).max(), "full hidden diff", abs(xy.cpu().numpy() - g["s_hidden"]).max())
    kc = m._rt[1]["k"].float().cpu().numpy(); vc = m._rt[1]["v"].float().cpu().numpy()
    print(" K diff", abs(kc[:, 0, :, :L] - g["s_k"]).max(), "V diff", abs(vc[:, 0, :, :L] - g["s_v"]).max())
    print(" kv_len", m._rt[1]["kv_len"].tolist(), "x_len", m._rt[1]["x_len"].tolist())
    hd = m.decode_hidden(1, T(g["d_x"][0]))
    torch.cuda.synchronize()
    print(" decode hidden diff", abs(hd.cpu().numpy() - g["d_hidden"][0]).max(), "k_new diff", abs(m._rt[1]["k"].float().cpu().numpy()[:, 0, :, L] - g["d_k_new"]).max(), "kv_len", m._rt[1]["kv_len"].tolist())
    # batch of two
    xs = [g["b0_x"], g["b1_x"]]; ys = [g["b0_y"], g["b1_y"]]; bs = [g["b0_bert"], g["b1_bert"]]
    xy, xl, yl, _, _ = m.embed_prompt([T(a) for a in xs], [T(a) for a in ys], [T(a) for a in bs])
    print(" batch embed diff", abs(xy.cpu().numpy() - g["b_xy"]).max())
    m.prefill(2, 0, xy, xl, yl); torch.cuda.synchronize()
    h = xy.cpu().numpy()
    for b in range(2):
        n = len(xs[b]) + len(ys[b]); print("  row", b, "hidden diff", abs(h[b, :n] - g["b_hidden"][b, :n]).max())

def t2s_infer(dtype, graph=True):
    gi = np.load(os.path.join(ROOT, "tests/golden/t2s_infer.npz"))
    cfg = synth.gpt_config()
    for name in "abc":
        seed, p, t, n = (int(v) for v in gi[name + "_cfg"])
        w = synth.gpt_weights(cfg, seed=seed, eos_gain=float(gi[name + "_eos_gain"]))
        cache = [tuple(int(v) for v in c) for c in gi[name + "_cache"]]
        m = Text2SemanticDecoder(cfg); m.load_state_dict(w); m.initialize_runtime(dtype, dev, cache); m.use_graph = graph
        x, y = gi[name + "_x"], gi[name + "_y"]
        t0 = time.time()
        tok = m.infer(T(x)[None], T(y)[None], torch.zeros(1, len(x), 1024, device=dev), top_k=1)
        torch.cuda.synchronize(); dt = time.time() - t0
        tok = tok[0, 0].cpu().numpy(); ref = gi[name + "_tokens"]
        nm = min(len(tok), len(ref)); eq = (tok[:nm] == ref[:nm]); first = int(np.argmin(eq)) if not eq.all() else -1
        print(dtype, "graph" if graph else "eager", name, "len", len(tok), len(ref), "match", float(eq.mean()), "first mismatch", first, "%.1f ms/tok" % (dt * 1000 / max(1, len(tok))))
        del m

def t2s_batched(dtype):
    g = np.load(os.path.join(ROOT, "tests/golden/t2s_batched.npz"))
    cfg = synth.gpt_config()
    for name in "rs":
        seed = int(g[name + "_seed"])
        w = synth.gpt_weights(cfg, seed=seed, eos_gain=float(g[name + "_eos_gain"]))
        cache = [tuple(int(v) for v in c) for c in g[name + "_cache"]]
        m = Text2SemanticDecoder(cfg); m.load_state_dict(w); m.initialize_runtime(dtype, dev, cache)
        rs = [synth.synth_request(200 + i, int(p), int(t), int(n), seed=seed) for i, (p, t, n) in enumerate(g[name + "_reqs"])]
        pred, orig = m.infer_batched([T(r[0]) for r in rs], [T(r[1]) for r in rs], [T(r[2]) for r in rs], top_k=1)
        print(dtype, name, "orig", orig.tolist(), g[name + "_orig"].tolist(), "lens", [len(p) for p in pred])
        for i, pt in enumerate(pred):
            ref = g["%s_tok%d" % (name, i)]; pt = pt.cpu().numpy(); nm = min(len(pt), len(ref))
            print("   ", i, len(pt), len(ref), "match", float((pt[:nm] == ref[:nm]).mean()) if nm else 1.0)
        del m

def vocoder(dtype):
    g = np.load(os.path.join(ROOT, "tests/golden/vocoder.npz"))
    for ver, Tn, tag in [("v2Pro", 50, "c"), ("v2Pro", 55, "pf"), ("v2ProPlus", 50, "c"), ("v2", 23, "c")]:
        hps = synth.sovits_hps(ver); w = synth.sovits_weights(hps, seed=int(g["seed"]), hot_path_only=True)
        v = _VocoderNative(hps["model"], {k: torch.from_numpy(a) for k, a in w.items()}, dtype, dev)
        name = "%s_T%d_%s" % (ver, Tn, tag)
        z, ge = T(g[name + "_z"]), T(g[name + "_ge"]); mask = torch.ones(1, 1, Tn, device=dev)
        zf = v.flow(z, mask, ge); torch.cuda.synchronize()
        print(dtype, name, "flow diff", abs(zf.cpu().numpy() - g[name + "_flow"]).max())
        od = v.dec(T(g[name + "_flow"]), ge); torch.cuda.synchronize()
        o = v.flow_dec(z, mask, ge); torch.cuda.synchronize()
        ref = g[name + "_o"]
        print("    dec-only diff", abs(od.cpu().numpy()[0, 0] - ref).max(), "flow_dec diff", abs(o.cpu().numpy()[0, 0] - ref).max(), "finite", bool(torch.isfinite(o).all()))
        del v

if __name__ == "__main__":
    which = sys.argv[1:] or ["layers", "infer", "batched", "vocoder"]
    for dt in (torch.float32, torch.bfloat16):
        for w_ in which:
            try:
                {"layers": t2s_layers, "infer": t2s_infer, "batched": t2s_batched, "vocoder": vocoder}[w_](dt)
            except Exception as e:
                import traceback; traceback.print_exc()
