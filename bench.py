#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement on MI355X (contract in the round prompt).

Workload (BASELINE.json configs[1]): "V2Pro bs=1 greedy AR decode + vocoder,
hipGraph on, bf16".  One STEP = one synthetic utterance through the hot path on each GPU:
  GPT:  100 phonemes (40 prompt + 60 target, zero BERT features) + 100 prompt semantic tokens,
        prefill then greedy decode of a FIXED 250 tokens (kv 200 -> 450; the reference API has
        no max-token argument, so the length is pinned by the bucket list [(1,256),(1,450)] and
        an EOS row of zero weight, i.e. EOS never wins);
  SoVITS: flow + Generator for those 250 tokens = 500 frames = 10 s of 32 kHz audio
        (z_p and ge synthetic; the text/ssl encoder enc_p is outside this timed hot path).
Weights are seeded random tensors of the real architecture (no checkpoints exist offline).

value = semantic tokens/s of the whole job end to end (AR + vocoder), all ranks; extra keys give
the AR-only rate, the vocoder audio-s/s, p50 TTFT, per-kernel rooflines and the CPU baseline
(the oracle = a C/OpenMP restatement of the reference's CPU path, timed on this box's cores).

N>1: one process per GPU (torch.distributed, backend nccl == RCCL), utterances are independent
(weak scaling: every rank runs its own utterance per step); the only collective is the broadcast
of the reference-speaker embedding `ge` from rank 0 at the start of each step.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_TFLOPS = 2500.0  # dense bf16 peak
MFMA_F32_TFLOPS = 157.3

N_PROMPT_PH, N_TEXT_PH, N_PROMPT_TOK, N_NEW = 40, 60, 100, 250
GPT_CACHE = [(1, 256), (1, 450)]
FRAMES = 2 * N_NEW


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--version", default="v2Pro", choices=["v2", "v2Pro", "v2ProPlus"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sampling / time-to-first-audio extras")
    ap.add_argument("--ttft-runs", type=int, default=50)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (CI on a 1-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    return ap.parse_args()


def vocoder_algorithmic(version, dtype_bytes):
    """layer-streaming conv I/O elements per 50 Hz frame, and FLOPs per frame."""
    elems = {"v2": 1655620 + 19994, "v2Pro": 1655620 + 19994, "v2ProPlus": 2483012 + 19994}[version]
    flops = {"v2": 813.1e6 + 14.2e6, "v2Pro": 813.1e6 + 14.2e6, "v2ProPlus": 1828.4e6 + 14.2e6}[version]
    return elems * dtype_bytes, flops


def _usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline_worker(version):
    """Runs in a child process (so a pathological host cannot hang the bench): the oracle (kind
    "port") on this box's host cores.  Thread count = the fastest of a short calibration sweep
    (a 1-row GEMV step does not scale to hundreds of threads), reported as `cores`."""
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    from gsv_tts_lite_amd import synth
    from oracle import oracle as orc
    cfg = synth.gpt_config()
    gw = synth.gpt_weights(cfg, seed=1234, eos_gain=0.0)
    hps = synth.sovits_hps(version)
    sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
    gin = hps["model"]["gin_channels"]
    x, y, bert, _ = synth.synth_request(0, N_PROMPT_PH, N_TEXT_PH, N_PROMPT_TOK, seed=1234)
    ge = synth.synth_ge(0, gin, 1234)
    z = synth.hashed_uniform("bench.z", (1, 192, FRAMES), 1234) * np.float32(1.2)
    avail = _usable_cores()
    o = orc.T2SOracle(cfg, gw, GPT_CACHE)
    xin = np.zeros((1, 512), np.float32)
    best, best_t = 1, 1e9
    for nt in [c for c in (4, 8, 16, 32, 64, 128, 256) if c <= avail] or [avail]:
        orc.set_num_threads(nt)
        o.decode(xin, 1, [200])
        t0 = time.perf_counter()
        for i in range(3):
            o.decode(xin, 1, [200 + i])
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = nt, dt
    orc.set_num_threads(best)
    t0 = time.perf_counter()
    tok = o.infer(x, y, bert, top_k=1)
    t_ar = time.perf_counter() - t0
    vo = orc.VocoderOracle(hps, sw)
    fs = 100
    t0 = time.perf_counter()
    vo.flow_dec(z[0, :, :fs], np.ones(fs, np.float32), ge[0])
    t_v = (time.perf_counter() - t0) * (FRAMES / fs)
    n = len(tok)
    print(json.dumps({
        "value": n / (t_ar + t_v), "unit": "semantic_tokens/s", "cores": best, "kind": "port",
        "sample": "oracle C/OpenMP fp32 on %d of %d usable host threads (best of a calibration sweep): full AR phase "
                  "(prefill + %d greedy tokens) once; flow+Generator on %d of %d frames scaled x%g"
                  % (best, avail, n, fs, FRAMES, FRAMES / fs),
        "ar_tokens_per_s": n / t_ar, "vocoder_audio_s_per_s": (FRAMES / 50.0) / t_v}))


def cpu_baseline(version, timeout_s=240):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--version", version],
                           capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            return json.loads(line[-1])
        return {"value": None, "unit": "semantic_tokens/s", "cores": 0, "kind": "port",
                "sample": "cpu baseline worker failed: %s" % r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "semantic_tokens/s", "cores": 0, "kind": "port",
                "sample": "cpu baseline worker exceeded %ds" % timeout_s}


def main():
    a = parse()
    if a.cpu_baseline_worker:
        cpu_baseline_worker(a.version)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.share_gpu:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)
    else:
        dist = None
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from gsv_tts_lite_amd import synth, _native as N
    from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
    from gsv_tts_lite_amd.sovits import _VocoderNative
    from gsv_tts_lite_amd import scheduler

    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    sbytes = 2 if a.dtype == "bf16" else 4
    cfg = synth.gpt_config()
    gw = synth.gpt_weights(cfg, seed=1234, eos_gain=0.0)
    hps = synth.sovits_hps(a.version)
    sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
    gin = hps["model"]["gin_channels"]

    t2s = Text2SemanticDecoder(cfg)
    t2s.load_state_dict(gw)
    t2s.initialize_runtime(dtype, dev, GPT_CACHE)
    t2s.use_graph = not a.no_graph
    voc = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, dtype, dev)

    # every rank owns a different utterance stream (weak scaling); the speaker embedding comes from rank 0
    def request(i):
        x, y, bert, _ = synth.synth_request(rank * 100003 + i, N_PROMPT_PH, N_TEXT_PH, N_PROMPT_TOK, seed=1234)
        return (torch.from_numpy(x)[None].to(dev), torch.from_numpy(y)[None].to(dev),
                torch.from_numpy(bert)[None].to(dev))
    reqs = [request(i) for i in range(max(1, min(8, a.steps + a.warmup)))]
    ge_src = torch.from_numpy(synth.synth_ge(0, gin, 1234)).to(dev)
    ge = ge_src.clone() if rank == 0 else torch.zeros_like(ge_src)
    z_np = synth.hashed_uniform("bench.z", (1, 192, FRAMES), 1234) * np.float32(1.2)
    z_p = torch.from_numpy(z_np).to(dev)
    mask = torch.ones(1, 1, FRAMES, device=dev)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    n_tok = [0]

    def step(i, timed_idx=None):
        scheduler.broadcast_speaker([ge], src=0)
        x, y, bert = reqs[i % len(reqs)]
        if timed_idx is not None:
            ev[timed_idx][0].record()
        tok = t2s.infer(x, y, bert, top_k=1)
        if timed_idx is not None:
            ev[timed_idx][1].record()
        audio = voc.flow_dec(z_p, mask, ge)
        if timed_idx is not None:
            ev[timed_idx][2].record()
        n_tok[0] = tok.shape[-1]
        return tok, audio

    log("models ready")
    for i in range(a.warmup):
        step(i)
    log("warmup done")
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i, i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = scheduler.max_over_ranks(time.perf_counter() - t0, device=dev)

    log("timed region done: %.3f s" % elapsed)
    tokens_per_step = n_tok[0]
    assert tokens_per_step == N_NEW, "expected %d tokens per utterance, got %d" % (N_NEW, tokens_per_step)
    t_ar = sum(e[0].elapsed_time(e[1]) for e in ev) / 1e3 / a.steps
    t_voc = sum(e[1].elapsed_time(e[2]) for e in ev) / 1e3 / a.steps
    audio_s = FRAMES / 50.0
    value = world * a.steps * tokens_per_step / elapsed

    out = {
        "metric": "semantic_tokens_per_sec_end_to_end (GPT AR incl. prefill + flow/Generator vocoder); RTF^-1 = value/25",
        "value": value, "unit": "semantic_tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic (seeded random weights of the real architecture, synthetic phoneme/token ids)",
        "config": {"workload": "configs[1]: %s bs=1 greedy AR decode (%d tokens, kv 200->450) + flow/Generator vocoder "
                               "(%d frames = %.0f s audio) per GPU per step, hipGraph %s" %
                               (a.version, N_NEW, FRAMES, audio_s, "off" if a.no_graph else "on"),
                   "utterances_per_step": world, "gpt_cache": GPT_CACHE, "parallelism": "replicas x%d, ge broadcast" % world},
        "audio_s_per_s_end_to_end": world * a.steps * audio_s / elapsed,
        "ar_tokens_per_s_per_gpu": tokens_per_step / t_ar,
        "vocoder_audio_s_per_s_per_gpu": audio_s / t_voc,
        "ar_ms_per_token": t_ar / tokens_per_step * 1e3,
        "vocoder_ms": t_voc * 1e3,
    }

    if rank == 0:
        # ---- p50 TTFT: prefill + first sample available on the host (ref-audio caches warm)
        x, y, bert = reqs[0]
        rt = t2s._rt[1]
        tt = []
        for _ in range(a.ttft_runs):
            torch.cuda.synchronize(dev)
            s0 = time.perf_counter()
            xy, xl, yl, _, _ = t2s.embed_prompt([x[0]], [y[0]], [bert[0]])
            t2s.prefill(1, 0, xy, xl, yl)
            t2s._flush(1)
            _ = int(rt["pre_tokens"][0, N_PROMPT_PH + N_TEXT_PH + N_PROMPT_TOK].item())
            tt.append((time.perf_counter() - s0) * 1e3)
        out["ttft_ms_p50"] = float(np.median(tt))
        log("ttft done")

        # ---- extras (not part of `value`): default-parameter sampling, and time to first audio
        # (first 25-token chunk + 50-frame vocoder pass)
        try:
            if a.no_extras:
                raise RuntimeError("--no-extras")
            for _ in range(2):
                torch.cuda.synchronize(dev); s0 = time.perf_counter()
                tk = t2s.infer(x, y, bert, top_k=15, repetition_penalty=1.35)
                torch.cuda.synchronize(dev); dt = time.perf_counter() - s0
            out["sampled_top_k15_ms_per_token"] = dt * 1e3 / max(1, int(tk.shape[-1]))
            z50, m50 = z_p[:, :, :50].contiguous(), mask[:, :, :50].contiguous()
            ta = []
            for _ in range(10):
                torch.cuda.synchronize(dev); s0 = time.perf_counter()
                xy, xl, yl, _, _ = t2s.embed_prompt([x[0]], [y[0]], [bert[0]])
                t2s.prefill(1, 0, xy, xl, yl)
                t2s._decode(1, 25)
                voc.flow_dec(z50, m50, ge)
                torch.cuda.synchronize(dev)
                ta.append((time.perf_counter() - s0) * 1e3)
            out["ttfa_ms_p50"] = float(np.median(ta))
            # first-batch TTFT at 32 slots: one packed prefill of 32 prompts + the first sample of each,
            # on a second decoder instance so that the timed bs=1 runtime above is not re-laid-out
            t32 = Text2SemanticDecoder(cfg)
            t32.load_state_dict(gw)
            t32.initialize_runtime(dtype, dev, [(32, 512)])
            r32 = [synth.synth_request(1000 + i, N_PROMPT_PH, N_TEXT_PH, N_PROMPT_TOK, seed=1234) for i in range(32)]
            xs = [torch.from_numpy(r[0]).to(dev) for r in r32]
            ys = [torch.from_numpy(r[1]).to(dev) for r in r32]
            bs_ = [torch.from_numpy(r[2]).to(dev) for r in r32]
            tb = []
            for _ in range(12):
                torch.cuda.synchronize(dev); s0 = time.perf_counter()
                xy, xl, yl, _, _ = t32.embed_prompt(xs, ys, bs_)
                t32.prefill(32, 0, xy, xl, yl)
                t32._flush(32)
                _ = t32._rt[32]["pre_tokens"][:, N_PROMPT_PH + N_TEXT_PH + N_PROMPT_TOK].cpu()
                tb.append((time.perf_counter() - s0) * 1e3)
            out["ttft_bs32_first_batch_ms_p50"] = float(np.median(tb[2:]))
            del t32
            # the vocoder as TTS.infer_batched feeds it (TTS.py:728-764): 10 utterances time-concatenated, per-frame ge
            T10 = 10 * FRAMES
            z10 = z_p.repeat(1, 1, 10).contiguous()
            m10 = torch.ones(1, 1, T10, device=dev)
            ge10 = ge.expand(-1, -1, T10).contiguous()
            for _ in range(2):
                voc.flow_dec(z10, m10, ge10)
            torch.cuda.synchronize(dev); s0 = time.perf_counter()
            for _ in range(3):
                voc.flow_dec(z10, m10, ge10)
            torch.cuda.synchronize(dev)
            t10 = (time.perf_counter() - s0) / 3
            vb10, vf10 = vocoder_algorithmic(a.version, sbytes)
            out["roofline_vocoder_batch10"] = {
                "bound": "hbm", "achieved": vb10 * T10 / t10 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": vb10 * T10 / t10 / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms_per_10s_audio": t10 * 1e3 / 10,
                "mfma_tflops": vf10 * T10 / t10 / 1e12,
                "note": "flow+Generator on 10 time-concatenated utterances (100 s of audio) in one pass, per-frame ge"}
        except Exception as exc:   # extras must never cost the bench line
            log("extras skipped: %r" % (exc,))

        # ---- roofline of the decode-step kernels (HIP events on the launch stream, live state: kv = 450)
        ms = (ctypes.c_float * 4)()
        N.check(N.lib().gsv_t2s_time_kernels(t2s._h, 1, 20, ms, N.current_stream_ptr(dev)))
        log("kernel timing done")
        kv = 450
        w_attn = (1536 * 512 + 512 * 512) * sbytes
        b_attn = w_attn + 2 * kv * 512 * sbytes + 2 * 512 * sbytes
        b_ffn = 2 * 2048 * 512 * sbytes
        b_log = 1025 * 512 * sbytes
        kern = []
        for name, t_ms, byts, per_tok in (("t2s_attn_kernel", ms[0], b_attn, 24), ("t2s_ffn_kernel", ms[1], b_ffn, 24),
                                          ("t2s_logits_kernel", ms[2], b_log, 1), ("t2s_token_kernel", ms[3], 4096, 1)):
            gbs = byts / (t_ms * 1e-3) / 1e9
            kern.append({"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": t_ms * 1e3,
                         "algorithmic_bytes_per_launch": byts, "launches_per_token": per_tok,
                         "us_per_token": t_ms * 1e3 * per_tok})
        dom = max(kern[:2], key=lambda k: k["us_per_token"])
        out["roofline"] = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        out["roofline"]["kernel"] = dom["kernel"]
        out["roofline"]["note"] = ("dominant kernel of the timed region (24 launches/token); algorithmic bytes = its weights "
                                   "(+ K/V rows at kv=450 for attn) per launch, SURVEY.md 8(d); bs=1 decode is latency-bound")
        out["roofline_kernels"] = kern
        vbytes, vflops = vocoder_algorithmic(a.version, sbytes)
        gbs = vbytes * FRAMES / t_voc / 1e9
        peak_tf = MFMA_BF16_TFLOPS if a.dtype == "bf16" else MFMA_F32_TFLOPS
        out["roofline_vocoder"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                   "mfma_tflops": vflops * FRAMES / t_voc / 1e12,
                                   "mfma_frac": vflops * FRAMES / t_voc / 1e12 / peak_tf,
                                   "note": "whole flow+Generator pass; algorithmic bytes = layer-streaming conv I/O, SURVEY.md 8(d)"}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tr = json.load(open(tfile))
                for k in out["roofline_kernels"]:
                    if k["kernel"] in tr:
                        k["traffic"] = tr[k["kernel"]]
                out["roofline"]["traffic"] = tr.get(out["roofline"]["kernel"])
                if a.version == "v2Pro" and a.dtype == "bf16":   # the PMC passes were taken on this configuration
                    out["roofline_vocoder"]["traffic"] = tr.get("vocoder_pass")
            except Exception:
                pass
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (subprocess) ...")
            cb = cpu_baseline(a.version)
            out["cpu_baseline"] = cb
            if cb.get("value"):
                out["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
