#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement on MI355X (contract in the round prompt).

--workload single (default; BASELINE.json configs[1]): "V2Pro bs=1 greedy AR decode + vocoder,
hipGraph on, bf16".  One STEP = one synthetic utterance through the hot path on each GPU:
  GPT:  100 phonemes (40 prompt + 60 target, zero BERT features) + 100 prompt semantic tokens, prefill then greedy
        decode of a FIXED 250 tokens (kv 200 -> 450; buckets [(1,512),(1,1024)] as SURVEY 8(d), the length pinned by
        infer(max_new_tokens=250) and an EOS row of zero weight, i.e. EOS never wins);
  SoVITS: flow + Generator for those 250 tokens = 500 frames = 10 s of 32 kHz audio (z_p and ge synthetic; the
        text/ssl encoder enc_p is outside this timed hot path).
--workload cb (configs[2] with --version v2ProPlus on one GPU, configs[3] = v2Pro on 8): continuous batching through the
multi-GPU engine (gsv_tts_lite_amd/engine.py): 256 mixed-length requests PER GPU through 32 slots per GPU, requests
dealt on demand from one shared cursor, then the flow + Generator over every finished utterance, time-concatenated in
batches of 10 as TTS.infer_batched feeds it.  One STEP = one pass over the whole queue.  --dtype fp8 --slots 64 is
configs[4] (e4m3 QKV / FFN operands in the batched decode step).

Weights are seeded random tensors of the real architecture (no checkpoints exist offline).
value = semantic tokens/s of the whole job end to end (AR + vocoder), all ranks; extra keys give the AR-only rate,
the vocoder audio-s/s, p50 TTFT, per-kernel and step-level rooflines, the fp32 parity-mode rates and the CPU baseline
(the oracle = a C/OpenMP restatement of the reference's CPU path, timed on this box's cores).

N>1: one process per GPU (torch.distributed, backend nccl == RCCL).  Utterances are independent; the only collective
is the ONE broadcast of the reference-speaker embedding `ge` from rank 0 when the speaker is first seen (setup, not
per step); `cb` additionally shares the request cursor through the process group's store.
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
GPT_CACHE = [(1, 512), (1, 1024)]
FRAMES = 2 * N_NEW
GPT_PARAMS = 76.02e6 + 0.16e6              # block linears + predict layer, + biases / LayerNorm
KV_BYTES_PER_POS = 2 * 24 * 512            # x dtype bytes: K and V rows of 24 layers
CB_REQUESTS_PER_GPU, CB_SLOTS = 256, 32


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="single", choices=["single", "cb"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--version", default="v2Pro", choices=["v2", "v2Pro", "v2ProPlus"])
    ap.add_argument("--slots", type=int, default=CB_SLOTS, help="cb: slots per GPU")
    ap.add_argument("--requests", type=int, default=CB_REQUESTS_PER_GPU, help="cb: requests per GPU per step")
    ap.add_argument("--sync-refill", action="store_true", help="cb: refill finished slots as the reference does, every slot waiting for "
                    "the prompt pass (default: the prompt pass runs on a side stream and the slot joins when it is done)")
    ap.add_argument("--lpt-budget", action="store_true", help="cb: order the queue by prompt length + the request's token budget (longest "
                    "first) instead of by text length alone; the default workload's budgets are independent of the text on purpose")
    ap.add_argument("--overlap", action="store_true", help="cb: vocoder batches on a side stream while the slot loop decodes (measured "
                    "+1 .. +4 %% end to end: the two share the chip) instead of after it in TTS.infer_batched's length-balanced order")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sampling / time-to-first-audio / fp32 extras")
    ap.add_argument("--no-cb32", action="store_true", help="single: skip the cb32 sub-record (32-slot continuous batching through the engine, "
                    "vocoder stage and rank-0 gather inside its timed steps; ~6 s)")
    ap.add_argument("--cb32-steps", type=int, default=2, help="single: timed passes over the 256-requests-per-GPU queue of the cb32 sub-record (after 1 warm-up pass)")
    ap.add_argument("--ttft-runs", type=int, default=50)
    ap.add_argument("--cpu-baseline-worker", default="", help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (CI on a 1-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--stub-decoder", action="store_true", help="cb, CPU ranks over gloo, NO kernels: the control path of the multi-GPU run (own "
                    "launcher, shared request cursor, token exchange, vocoder batches dealt over the ranks, rank-0 gather, max over ranks) with a "
                    "stub slot loop and a stub vocoder; its line says so and is not a measurement (tests/test_bench_world8_gloo.py)")
    ap.add_argument("--stub-die-rank", type=int, default=-1, help=argparse.SUPPRESS)   # that rank exits before its first collective (launcher test)
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


def _calibrate_threads(o, orc):
    """fastest OpenMP team for a 1-row decode step (a GEMV does not scale to hundreds of threads)"""
    avail = _usable_cores()
    xin = np.zeros((1, 512), np.float32)
    best, best_t, sweep = 1, 1e9, {}
    for nt in sorted(set([c for c in (1, 4, 8, 16, 32, 64, 128, 256) if c <= avail] + [avail])):
        orc.set_num_threads(nt)
        o.decode(xin, 1, [200])
        t0 = time.perf_counter()
        for i in range(3):
            o.decode(xin, 1, [200 + i])
        dt = time.perf_counter() - t0
        sweep[str(nt)] = round(dt / 3 * 1e3, 3)
        if dt < best_t:
            best, best_t = nt, dt
    orc.set_num_threads(best)
    return best, avail, sweep


def cpu_baseline_worker(kind, version):
    """Runs in a child process (so a pathological host cannot hang the bench): the oracle (kind "port") on this box's
    host cores, on a bounded sample of the same workload."""
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    from gsv_tts_lite_amd import synth
    from oracle import oracle as orc
    cfg = synth.gpt_config()
    hps = synth.sovits_hps(version)
    sw = synth.sovits_weights(hps, seed=1234, hot_path_only=True)
    gin = hps["model"]["gin_channels"]
    ge = synth.synth_ge(0, gin, 1234)
    vo = orc.VocoderOracle(hps, sw)
    if kind == "single":
        gw = synth.gpt_weights(cfg, seed=1234, eos_gain=0.0)
        x, y, bert, _ = synth.synth_request(0, N_PROMPT_PH, N_TEXT_PH, N_PROMPT_TOK, seed=1234)
        z = synth.hashed_uniform("bench.z", (1, 192, FRAMES), 1234) * np.float32(1.2)
        o = orc.T2SOracle(cfg, gw, [(1, 256), (1, 450)])       # 250 tokens: the cache fills at kv 450
        best, avail, sweep = _calibrate_threads(o, orc)
        t0 = time.perf_counter()
        tok = o.infer(x, y, bert, top_k=1)
        t_ar = time.perf_counter() - t0
        orc.set_num_threads(min(avail, max(best, 16)))
        t0 = time.perf_counter()
        vo.flow_dec(z[0], np.ones(FRAMES, np.float32), ge[0])
        t_v = time.perf_counter() - t0
        n = len(tok)
        print(json.dumps({
            "value": n / (t_ar + t_v), "unit": "semantic_tokens/s", "cores": best, "kind": "port",
            "sample": "oracle C/OpenMP fp32, %d of %d usable host threads for the AR phase (best of a calibration sweep), "
                      "%d for the vocoder: one whole utterance = prefill + %d greedy tokens + flow/Generator on all %d frames "
                      "(no enc_p leg: the oracle restates the hot path, not the text / ssl encoder)"
                      % (best, avail, min(avail, max(best, 16)), n, FRAMES),
            "ar_tokens_per_s": n / t_ar, "vocoder_audio_s_per_s": (FRAMES / 50.0) / t_v,
            "decode_step_ms_by_threads": sweep, "usable_threads": avail,
            "note": "decode_step_ms_by_threads = one 24-layer decode step (kv 200) of the oracle per OpenMP team size on this host, the sweep "
                    "`cores` was picked from (a 1-row GEMV does not scale to every core); the all-threads figure is its last entry"}))
    else:
        gw = synth.gpt_weights(cfg, seed=1234, eos_gain=4.0)
        nreq, slots = 12, 8
        lens = synth.mixed_lengths(nreq)
        reqs = [synth.synth_request(i, 40, t, n) for i, (t, n) in enumerate(lens)]
        o = orc.T2SOracle(cfg, gw, [(slots, 256), (slots, 400)])   # EOS- / capacity-terminated: ~ the GPU run's token mix
        avail = _usable_cores()
        orc.set_num_threads(min(avail, 32))
        t0 = time.perf_counter()
        pred, _ = o.infer_batched([r[0] for r in reqs], [r[1] for r in reqs], [r[2] for r in reqs], top_k=1)
        t_ar = time.perf_counter() - t0
        ntok = int(sum(len(p) for p in pred))
        fs = min(2 * len(pred[0]), 200)
        z = synth.hashed_uniform("bench.z", (1, 192, fs), 1234) * np.float32(1.2)
        t0 = time.perf_counter()
        vo.flow_dec(z[0], np.ones(fs, np.float32), ge[0])
        t_v = (time.perf_counter() - t0) * (2 * ntok / fs)
        print(json.dumps({
            "value": ntok / (t_ar + t_v), "unit": "semantic_tokens/s", "cores": min(avail, 32), "kind": "port",
            "sample": "oracle C/OpenMP fp32 on %d host threads: continuous batching of %d mixed-length requests through %d "
                      "slots (%d tokens); vocoder = flow/Generator timed on %d frames, scaled to the %d frames generated"
                      % (min(avail, 32), nreq, slots, ntok, fs, 2 * ntok),
            "ar_tokens_per_s": ntok / t_ar, "vocoder_audio_s_per_s": (2 * ntok / 50.0) / t_v}))


def cpu_baseline(kind, version, timeout_s=300):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", kind, "--version", version],
                           capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            return json.loads(line[-1])
        return {"value": None, "unit": "semantic_tokens/s", "cores": 0, "kind": "port",
                "sample": "cpu baseline worker failed: %s" % r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "semantic_tokens/s", "cores": 0, "kind": "port",
                "sample": "cpu baseline worker exceeded %ds" % timeout_s}


def _torch_dtype(name):
    return {"bf16": torch.bfloat16, "fp32": torch.float32, "fp8": torch.float8_e4m3fn}[name]


def csrc_sha16():
    """sha256 over the kernel sources (csrc/*.h, *.hip, sorted by name): what a PMC pass in profiles/ is valid for.  The GPU box
    has no .git, so the stamp is content-based; profiles/traffic.json carries the same hash plus the commit it was taken at."""
    import hashlib
    d = os.path.join(ROOT, "gsv-tts-lite_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _step_traffic(slots, dtype_name):
    """fabric bytes of ONE batched decode step (32 sequences, bf16) from the committed PMC passes, or None when the kernels changed since
    (profiles/traffic.json "batched_step_b32": tools/pmc_traffic.py, a pass of tools/step_time.py 32 bf16)"""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if slots != 32 or dtype_name != "bf16" or tr.get("_meta", {}).get("csrc_sha16") != csrc_sha16():
            return None
        return tr.get("batched_step_b32")
    except Exception:  # noqa: BLE001
        return None


def _profile_traffic(out, a):
    """HBM traffic per launch from the committed PMC passes (profiles/traffic.json): counters cannot be read in-process.  The file
    is stamped with the commit and the csrc hash it was measured at; when the kernels changed since, traffic stays null."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return
    try:
        tr = json.load(open(tfile))
        meta = tr.get("_meta", {})
        out["roofline"]["traffic_commit"] = meta.get("commit")
        out["roofline"]["traffic_csrc_sha16"] = meta.get("csrc_sha16")
        if meta.get("csrc_sha16") != csrc_sha16():
            out["roofline"]["traffic_source"] = ("profiles/traffic.json was measured at commit %s (csrc %s); the kernels changed since (csrc %s): "
                                                 "no traffic figure is claimed for this build" % (meta.get("commit"), meta.get("csrc_sha16"), csrc_sha16()))
            return
        for k in out.get("roofline_kernels", []):
            if k["kernel"] in tr:
                k["traffic"] = tr[k["kernel"]]
        if "kernel" in out["roofline"]:
            out["roofline"]["traffic"] = tr.get(out["roofline"]["kernel"])
            out["roofline"]["traffic_source"] = ("profiles/traffic.json: per-launch mean of two rocprofv3 --pmc passes (FETCH_SIZE x2 per the "
                                                 "micro-architecture guide, WRITE_SIZE) over this command with --steps 2 --no-extras --no-cb32, "
                                                 "tools/sweep.sh at the commit above; hardware counters cannot be read from inside the process")
        if a.version == "v2Pro" and a.dtype == "bf16" and "roofline_vocoder" in out:
            out["roofline_vocoder"]["traffic"] = tr.get("vocoder_pass")
    except Exception:
        pass


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: this process starts the N ranks itself, one per GPU -- RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in each child's environment, device affinity = LOCAL_RANK (setup_dist selects
    cuda:LOCAL_RANK and checks it; every rank keeps ALL devices visible: RCCL's xGMI peer paths want to see the peers, so no
    HIP_VISIBLE_DEVICES mask) -- passes their output through and WATCHES them: a rank that dies takes the others down with a message,
    instead of leaving them in a collective or on the store cursor until a timeout.  Returns the exit code."""
    import socket
    import subprocess
    if not (a.share_gpu or a.stub_decoder) and torch.cuda.device_count() < a.gpus:
        raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s)" % (a.gpus, torch.cuda.device_count()))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", "4")
    base.update({"WORLD_SIZE": str(a.gpus), "LOCAL_WORLD_SIZE": str(a.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    log("launching %d ranks (own launcher, 127.0.0.1:%d): %s" % (a.gpus, port, " ".join(cmd[1:])))
    procs = []
    for r in range(a.gpus):
        env = dict(base)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "GROUP_RANK": "0"})
        procs.append(subprocess.Popen(cmd, env=env))
    rc, dead = 0, None
    try:
        while any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                c = p.poll()
                if c is not None and c != 0 and dead is None:
                    dead, rc = r, c
            if dead is not None:
                break
            time.sleep(0.05)
        if dead is None:
            for r, p in enumerate(procs):
                if p.returncode != 0 and rc == 0:
                    dead, rc = r, p.returncode
    finally:
        if dead is not None:
            alive = [r for r, p in enumerate(procs) if p.poll() is None]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            print("[bench] rank %d exited with code %d; the other ranks%s were stopped (they would have waited for it in a collective "
                  "or on the shared request cursor)" % (dead, rc, " " + str(alive) if alive else ""), file=sys.stderr, flush=True)
    return rc if rc != 0 else 0


def setup_dist(a):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if a.stub_decoder:
        if a.workload != "cb" or a.dist_backend != "gloo":
            raise SystemExit("--stub-decoder is the CPU control-path run: --workload cb --dist-backend gloo")
        if rank == a.stub_die_rank:
            sys.exit(3)
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
        return world, rank, torch.device("cpu"), dist
    if a.share_gpu and a.dist_backend == "nccl" and world > 1:
        raise SystemExit("--share-gpu needs --dist-backend gloo: RCCL wants one device per rank")
    if not a.share_gpu and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        # under a launcher too (the bare launch checks before it starts ranks): N ranks never share fewer than N devices silently
        raise SystemExit("bench.py --gpus %d: rank %d sees %d GPU(s) for %s ranks on this node; --share-gpu (with --dist-backend gloo) is the "
                         "explicit testing mode" % (a.gpus, rank, torch.cuda.device_count(), os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world != a.gpus:      # a line that says n_gpus: 1 for a --gpus 8 request is worse than no line
        raise SystemExit("bench.py --gpus %d is running with WORLD_SIZE=%d: launch it bare (it starts its own ranks) or under "
                         "torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    dist = None
    # GSV_FORCE_COLLECTIVES=1: a one-rank run still initialises the process group and sends every exchange through the backend's
    # collectives (engine.py) -- how a 1-GPU box proves that RCCL comes up under this exact command (tests/test_hip_rccl.py)
    if world > 1 or os.environ.get("GSV_FORCE_COLLECTIVES") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
        torch.cuda.set_device(local)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if not a.share_gpu and torch.cuda.current_device() != local:      # one process per GPU, the GPU its LOCAL_RANK names
        raise SystemExit("rank %d: LOCAL_RANK %d but the current device is %d" % (rank, local, torch.cuda.current_device()))
    return world, rank, dev, dist


def dist_info(a, world, book):
    import torch.distributed as td
    return {"world_size": world, "backend": (td.get_backend() + (" (RCCL)" if td.get_backend() == "nccl" else "")) if td.is_initialized() else None,
            "speaker_broadcasts": book.broadcasts, "ranks_share_one_gpu": bool(a.share_gpu)}


def timed_region(steps, warmup, step_fn, dev, dist):
    from gsv_tts_lite_amd import scheduler
    for i in range(warmup):
        step_fn(i, None)
    log("warmup done")
    _sync(dev)
    if dist is not None:
        dist.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(warmup + i, i)
    _sync(dev)
    if dist is not None:
        dist.barrier()
    _sync(dev)
    return scheduler.max_over_ranks(time.perf_counter() - t0, device=dev)


# ================================================================================================ configs[1]
def run_single(a):
    world, rank, dev, dist = setup_dist(a)
    from gsv_tts_lite_amd import synth, _native as N, engine
    from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
    from gsv_tts_lite_amd.sovits import _VocoderNative

    if a.dtype == "fp8":
        raise SystemExit("fp8 operands exist in the batched decode step only: use --workload cb --dtype fp8")
    dtype = _torch_dtype(a.dtype)
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

    # every rank owns a different utterance stream (weak scaling); the speaker embedding exists on rank 0 and is
    # broadcast ONCE, when the speaker is first seen (engine.SpeakerBook), not per utterance
    def request(i):
        x, y, bert, _ = synth.synth_request(rank * 100003 + i, N_PROMPT_PH, N_TEXT_PH, N_PROMPT_TOK, seed=1234)
        return (torch.from_numpy(x)[None].to(dev), torch.from_numpy(y)[None].to(dev), torch.from_numpy(bert)[None].to(dev))
    reqs = [request(i) for i in range(max(1, min(8, a.steps + a.warmup)))]
    book = engine.SpeakerBook(dev)
    ge = book.sync("speaker-0", [torch.from_numpy(synth.synth_ge(0, gin, 1234))] if rank == 0 else None)[0]
    z_np = synth.hashed_uniform("bench.z", (1, 192, FRAMES), 1234) * np.float32(1.2)
    z_p = torch.from_numpy(z_np).to(dev)
    mask = torch.ones(1, 1, FRAMES, device=dev)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    n_tok = [0]

    def step(i, timed_idx):
        x, y, bert = reqs[i % len(reqs)]
        if timed_idx is not None:
            ev[timed_idx][0].record()
        tok = t2s.infer(x, y, bert, top_k=1, max_new_tokens=N_NEW)
        if timed_idx is not None:
            ev[timed_idx][1].record()
        audio = voc.flow_dec(z_p, mask, ge)
        if timed_idx is not None:
            ev[timed_idx][2].record()
        n_tok[0] = tok.shape[-1]
        return tok, audio

    log("models ready")
    elapsed = timed_region(a.steps, a.warmup, step, dev, dist)
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
                   "utterances_per_step": world, "gpt_cache": GPT_CACHE, "max_new_tokens": N_NEW,
                   "parallelism": "replicas x%d, ge broadcast once per speaker (%d tensor broadcasts in this run)" % (world, book.broadcasts)},
        "dist": dist_info(a, world, book),
        "audio_s_per_s_end_to_end": world * a.steps * audio_s / elapsed,
        "ar_tokens_per_s_per_gpu": tokens_per_step / t_ar,
        "vocoder_audio_s_per_s_per_gpu": audio_s / t_voc,
        "ar_ms_per_token": t_ar / tokens_per_step * 1e3,
        "vocoder_ms": t_voc * 1e3,
    }

    # ---- cb32: BASELINE's "bs=1/32" second half, on every line.  v2Pro, 32 slots and 256 mixed-length requests per GPU through
    # ContinuousBatchingEngine (shared request cursor, token exchange, vocoder stage dealt over the ranks, every request's
    # samples gathered on rank 0 INSIDE the timed step); at --gpus 8 this record is configs[3].  Every rank takes part.
    cb32 = None
    if not a.no_cb32:
        try:
            w = CBWorkload(a, world, rank, dev, dist, a.version, a.dtype, CB_SLOTS, CB_REQUESTS_PER_GPU, voc=voc, book=book, ge=ge)
            el = timed_region(a.cb32_steps, 1, w.step, dev, dist)
            cb32 = w.record(el, a.cb32_steps, 1)
            if rank == 0:
                # BASELINE's "p50 TTFT ... bs=32": the first 32 requests of the queue arrive together -> ONE packed prompt pass of all of
                # them + the first decode step (every request's first token exists); outside the timed steps
                try:
                    first = list(range(CB_SLOTS))
                    ts = []
                    for _ in range(9):
                        _sync(dev); q0 = time.perf_counter()
                        with torch.inference_mode():
                            xy_, xl_, yl_, _, _ = w.t2s.embed_prompt([w.xs[c] for c in first], [w.ys[c] for c in first], [w.bs[c] for c in first])
                            w.t2s.prefill(CB_SLOTS, 0, xy_, xl_, yl_)
                            w.t2s._decode(CB_SLOTS, 1)
                        _sync(dev); ts.append(time.perf_counter() - q0)
                    cb32["ttft_ms_p50_first_batch"] = sorted(ts[2:])[len(ts[2:]) // 2] * 1e3
                    cb32["ttft_first_batch_note"] = "%d prompts (%d positions in all) in one packed prompt pass + the first decode step" % (
                        len(first), int(sum(int(w.xs[c].shape[0]) + int(w.ys[c].shape[0]) for c in first)))
                except Exception as exc:  # noqa: BLE001
                    log("cb32 first-batch TTFT failed: %r" % (exc,))
            for k in ("metric", "unit", "higher_is_better", "scaling", "vs_baseline", "data"):
                cb32.pop(k, None)
            log("cb32 done: %.0f tok/s" % cb32["value"])
            del w
            torch.cuda.empty_cache()
        except Exception as exc:
            if world > 1:
                raise          # a rank that drops out of a collective must not leave the others waiting
            log("cb32 failed: %r" % (exc,))
            cb32 = {"value": None, "error": repr(exc)}
    if cb32 is not None:
        out["cb32"] = cb32

    if rank == 0:
        x, y, bert = reqs[0]
        rt = t2s._rt[1]
        # ---- p50 TTFT: prefill + first sample available on the host (ref-audio caches warm)
        tt = []
        for _ in range(a.ttft_runs):
            _sync(dev)
            s0 = time.perf_counter()
            xy, xl, yl, _, _ = t2s.embed_prompt([x[0]], [y[0]], [bert[0]])
            t2s.prefill(1, 0, xy, xl, yl)
            t2s._flush(1)
            _ = int(rt["pre_tokens"][0, N_PROMPT_PH + N_TEXT_PH + N_PROMPT_TOK].item())
            tt.append((time.perf_counter() - s0) * 1e3)
        out["ttft_ms_p50"] = float(np.median(tt))
        log("ttft done")

        # ---- step-level roofline of the timed AR phase: algorithmic bytes per token = weights + the K/V rows read
        kv_avg = N_PROMPT_PH + N_TEXT_PH + N_PROMPT_TOK + (N_NEW - 1) / 2.0
        step_bytes = GPT_PARAMS * sbytes + KV_BYTES_PER_POS * sbytes * kv_avg
        ar_decode_s = t_ar - out["ttft_ms_p50"] * 1e-3 if t_ar > out["ttft_ms_p50"] * 1e-3 else t_ar
        gbs = step_bytes / (ar_decode_s / tokens_per_step) / 1e9
        out["roofline_step"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "traffic": None, "bytes_per_token": step_bytes, "ms_per_token": ar_decode_s / tokens_per_step * 1e3,
                                "note": "whole decode step (49 launches replayed from one hipGraph: the token kernel's work is in layer 0's attention kernel; t2s_token_kernel runs once per 5-step window as the flush): (76.2 M weights + K/V rows at the run's "
                                        "mean kv) x dtype bytes / measured time per token (AR phase minus the p50 prefill)"}

        # ---- roofline of the decode-step kernels: each class's 24 launches replayed from a hipGraph between two
        # hipEvents on the launch stream (no host launch cost; the dependent-launch gap of a real step is included)
        ms = (ctypes.c_float * 4)()
        N.check(N.lib().gsv_t2s_time_kernels(t2s._h, 1, 20, ms, N.current_stream_ptr(dev)))
        log("kernel timing done")
        kv = int(rt["kv_len"][0].item())
        w_attn = (1536 * 512 + 512 * 512) * sbytes
        b_attn = w_attn + 2 * kv * 512 * sbytes + 2 * 512 * sbytes
        b_ffn = 2 * 2048 * 512 * sbytes
        b_log = 1025 * 512 * sbytes
        kern = []
        for name, t_ms, byts, per_tok in (("t2s_attn_kernel", ms[0], b_attn, 24), ("t2s_ffn_kernel", ms[1], b_ffn, 24),
                                          ("t2s_logits_kernel", ms[2], b_log, 1), ("t2s_token_kernel", ms[3], 4096, 0.2)):
            g = byts / (t_ms * 1e-3) / 1e9
            kern.append({"kernel": name, "bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": g / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": t_ms * 1e3,
                         "algorithmic_bytes_per_launch": byts, "launches_per_token": per_tok, "us_per_token": t_ms * 1e3 * per_tok,
                         "kv_at_measurement": kv})
        dom = max(kern[:2], key=lambda k: k["us_per_token"])
        out["roofline"] = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        out["roofline"]["kernel"] = dom["kernel"]
        out["roofline"]["avg_launch_us"] = dom["avg_launch_us"]
        out["roofline"]["note"] = ("dominant kernel of the timed region (24 launches/token); algorithmic bytes = its weights (+ K/V rows "
                                   "at the live kv for attn) per launch, SURVEY.md 8(d); time = hipGraph replay of the class between "
                                   "hipEvents, which includes the dependent-launch gap (rocprofv3's per-dispatch figure in profiles/ "
                                   "excludes it); bs=1 decode is latency-bound")
        out["roofline_kernels"] = kern
        vbytes, vflops = vocoder_algorithmic(a.version, sbytes)
        g = vbytes * FRAMES / t_voc / 1e9
        peak_tf = MFMA_BF16_TFLOPS if a.dtype == "bf16" else MFMA_F32_TFLOPS
        out["roofline_vocoder"] = {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                                   "traffic": None, "mfma_tflops": vflops * FRAMES / t_voc / 1e12,
                                   "mfma_frac": vflops * FRAMES / t_voc / 1e12 / peak_tf,
                                   "note": "whole flow+Generator pass; algorithmic bytes = layer-streaming conv I/O, SURVEY.md 8(d)"}
        _profile_traffic(out, a)

        # ---- extras (not part of `value`)
        try:
            if a.no_extras:
                raise RuntimeError("--no-extras")
            for _ in range(2):   # default-parameter sampling
                _sync(dev); s0 = time.perf_counter()
                tk = t2s.infer(x, y, bert, top_k=15, repetition_penalty=1.35, max_new_tokens=N_NEW)
                _sync(dev); dt = time.perf_counter() - s0
            out["sampled_top_k15_ms_per_token"] = dt * 1e3 / max(1, int(tk.shape[-1]))
            z50, m50 = z_p[:, :, :50].contiguous(), mask[:, :, :50].contiguous()
            ta = []
            for _ in range(10):   # time to first audio: prefill + first 25-token chunk + 50-frame vocoder pass
                _sync(dev); s0 = time.perf_counter()
                xy, xl, yl, _, _ = t2s.embed_prompt([x[0]], [y[0]], [bert[0]])
                t2s.prefill(1, 0, xy, xl, yl)
                t2s._decode(1, 25)
                voc.flow_dec(z50, m50, ge)
                _sync(dev)
                ta.append((time.perf_counter() - s0) * 1e3)
            out["ttfa_ms_p50"] = float(np.median(ta))
            # end to end as SURVEY 8(d) words it (GPT + enc_p + flow_dec): the utterance's OWN tokens through SynthesizerTrn.decode
            # -- quantizer lookup, device enc_p, noise draw, flow + Generator in one library call -- instead of a synthetic z_p
            from gsv_tts_lite_amd.sovits import SynthesizerTrn
            vq = SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
            vq.load_state_dict(synth.sovits_weights(hps, seed=1234))
            vq.initialize_runtime(dtype, dev, [])
            txt = torch.from_numpy(synth.synth_request(rank * 100003, N_PROMPT_PH, N_TEXT_PH, N_PROMPT_TOK, seed=1234)[3])[None].to(dev)
            for _ in range(2):
                vq.decode(t2s.infer(x, y, bert, top_k=1, max_new_tokens=N_NEW), txt, ge, noise_scale=0.5)
            _sync(dev); s0 = time.perf_counter()
            nrep, t_dec = 5, 0.0
            for _ in range(nrep):
                tk = t2s.infer(x, y, bert, top_k=1, max_new_tokens=N_NEW)
                _sync(dev); s1 = time.perf_counter()
                au, _ = vq.decode(tk, txt, ge, noise_scale=0.5)
                _sync(dev); t_dec += time.perf_counter() - s1
            s2 = time.perf_counter()
            ta = []
            for _ in range(10):   # time to first audio through the REAL first-chunk path: prefill + 25 tokens + decode() of them (50 frames)
                _sync(dev); q0 = time.perf_counter()
                xy, xl, yl, _, _ = t2s.embed_prompt([x[0]], [y[0]], [bert[0]])
                t2s.prefill(1, 0, xy, xl, yl)
                t2s._decode(1, 25)
                vq.decode(tk[:, :, :25], txt, ge, noise_scale=0.5)
                _sync(dev)
                ta.append((time.perf_counter() - q0) * 1e3)
            out["ttfa_decode_ms_p50"] = float(np.median(ta))
            out["value_with_enc_p"] = nrep * N_NEW / (s2 - s0)
            out["decode_ms"] = t_dec / nrep * 1e3
            out["value_with_enc_p_note"] = ("semantic tokens/s of GPT + SynthesizerTrn.decode (enc_p + noise + flow + Generator, noise_scale 0.5) on "
                                            "the utterance's own %d tokens, %d utterances; `value` feeds the vocoder a synthetic z_p" % (N_NEW, nrep))
            del vq
            # the vocoder as TTS.infer_batched feeds it (TTS.py:728-764): 10 utterances time-concatenated, per-frame ge
            T10 = 10 * FRAMES
            z10 = z_p.repeat(1, 1, 10).contiguous()
            m10 = torch.ones(1, 1, T10, device=dev)
            ge10 = ge.expand(-1, -1, T10).contiguous()
            for _ in range(2):
                voc.flow_dec(z10, m10, ge10)
            _sync(dev); s0 = time.perf_counter()
            for _ in range(3):
                voc.flow_dec(z10, m10, ge10)
            _sync(dev)
            t10 = (time.perf_counter() - s0) / 3
            vb10, vf10 = vocoder_algorithmic(a.version, sbytes)
            out["roofline_vocoder_batch10"] = {
                "bound": "hbm", "achieved": vb10 * T10 / t10 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": vb10 * T10 / t10 / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms_per_10s_audio": t10 * 1e3 / 10,
                "mfma_tflops": vf10 * T10 / t10 / 1e12,
                "note": "flow+Generator on 10 time-concatenated utterances (100 s of audio) in one pass, per-frame ge"}
            # the bit-exact configuration (fp32 weights / KV / activations: the parity mode of tests/) measured as well
            if a.dtype != "fp32":
                del z10, m10, ge10
                t2f = Text2SemanticDecoder(cfg)
                t2f.load_state_dict(gw)
                t2f.initialize_runtime(torch.float32, dev, GPT_CACHE)
                vof = _VocoderNative(hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, torch.float32, dev)
                tok32 = t2f.infer(x, y, bert, top_k=1, max_new_tokens=N_NEW); au32 = vof.flow_dec(z_p, mask, ge)
                _sync(dev); s0 = time.perf_counter()
                for _ in range(3):
                    t2f.infer(x, y, bert, top_k=1, max_new_tokens=N_NEW)
                _sync(dev); t1 = time.perf_counter()
                for _ in range(3):
                    vof.flow_dec(z_p, mask, ge)
                _sync(dev); t2 = time.perf_counter()
                # where the fp32 AR time goes beyond the raw steps: its own prompt pass + first step (TTFT), the raw hipGraph step
                tf = []
                for _ in range(10):
                    _sync(dev); q0 = time.perf_counter()
                    xy, xl, yl, _, _ = t2f.embed_prompt([x[0]], [y[0]], [bert[0]])
                    t2f.prefill(1, 0, xy, xl, yl)
                    t2f._decode(1, 1)
                    _sync(dev)
                    tf.append((time.perf_counter() - q0) * 1e3)
                t2f._decode(1, 20); _sync(dev); q0 = time.perf_counter()
                t2f._decode(1, 100); _sync(dev)
                raw32 = (time.perf_counter() - q0) * 1e3 / 100
                ar32 = (t1 - s0) / 3 * 1e3
                out["fp32_parity_mode"] = {
                    "ar_tokens_per_s": 3 * N_NEW / (t1 - s0), "ar_ms_per_token": (t1 - s0) / 3 / N_NEW * 1e3,
                    "vocoder_ms": (t2 - t1) / 3 * 1e3, "tokens_per_s_end_to_end": 3 * N_NEW / (t2 - s0),
                    "ttft_ms_p50": float(np.median(tf)), "raw_step_ms": raw32,
                    "ar_ms_breakdown": {"utterance": ar32, "prompt_pass_and_first_step": float(np.median(tf)), "raw_steps": raw32 * (N_NEW - 1),
                                        "loop_and_readback": ar32 - float(np.median(tf)) - raw32 * (N_NEW - 1)},
                    "note": "same workload with dtype fp32: the configuration whose greedy tokens are bit-exact and whose waveform "
                            "is within 1e-3 of the fp32 CPU reference (tests/test_hip_t2s.py, test_hip_vocoder.py)"}
                # ---- precision ledger: what the benchmarked bf16 mode costs against the fp32 mode, from the product's own two modes
                # (no oracle here): the bench requests' greedy tokens and the bench waveform
                led = {"requests": []}
                for j in range(len(reqs)):
                    xj, yj, bj = reqs[j]
                    tb = t2s.infer(xj, yj, bj, top_k=1, max_new_tokens=N_NEW)[0, 0].cpu().numpy()
                    tf32 = (tok32 if j == 0 and xj is x else t2f.infer(xj, yj, bj, top_k=1, max_new_tokens=N_NEW))[0, 0].cpu().numpy()
                    n = min(len(tb), len(tf32))
                    neq = np.nonzero(tb[:n] != tf32[:n])[0]
                    led["requests"].append({"tokens": int(n), "matched_prefix": int(neq[0]) if neq.size else int(n),
                                            "agreement": float((tb[:n] == tf32[:n]).mean())})
                ab = voc.flow_dec(z_p, mask, ge).float().cpu().numpy().ravel()
                af = au32.float().cpu().numpy().ravel()
                led["matched_prefix_min"] = min(r["matched_prefix"] for r in led["requests"])
                led["matched_prefix_mean"] = float(np.mean([r["matched_prefix"] for r in led["requests"]]))
                led["token_agreement_mean"] = float(np.mean([r["agreement"] for r in led["requests"]]))
                led["waveform_max_abs_diff"] = float(np.abs(ab - af).max())
                led["waveform_mean_abs_diff"] = float(np.abs(ab - af).mean())
                led["waveform_rms_fp32"] = float(np.sqrt((af.astype(np.float64) ** 2).mean()))
                led["note"] = ("bf16-mode output against the fp32-mode output of the SAME library on the bench inputs (greedy, %d tokens per request; the "
                               "%d-frame waveform from the same z_p): once a token differs the two sequences are different utterances, so `agreement` "
                               "after the matched prefix measures decorrelation, not error" % (N_NEW, FRAMES))
                out["bf16_vs_fp32"] = led
                del t2f, vof
        except Exception as exc:   # extras must never cost the bench line
            log("extras skipped: %r" % (exc,))

        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (subprocess) ...")
            cb = cpu_baseline("single", a.version)
            out["cpu_baseline"] = cb
            if cb.get("value"):
                out["speedup_vs_cpu_baseline"] = value / cb["value"]
                ratios = {"end_to_end": value / cb["value"]}
                if cb.get("ar_tokens_per_s"):
                    ratios["ar_only"] = out["ar_tokens_per_s_per_gpu"] / cb["ar_tokens_per_s"]
                if cb.get("vocoder_audio_s_per_s"):
                    ratios["vocoder_only"] = out["vocoder_audio_s_per_s_per_gpu"] / cb["vocoder_audio_s_per_s"]
                f32 = out.get("fp32_parity_mode")
                if f32:
                    ratios["fp32_mode_end_to_end"] = f32["tokens_per_s_end_to_end"] / cb["value"]
                    if cb.get("ar_tokens_per_s"):
                        ratios["fp32_mode_ar_only"] = f32["ar_tokens_per_s"] / cb["ar_tokens_per_s"]
                out["speedup_vs_cpu_baseline_by_leg"] = ratios
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ================================================================================================ configs[2] / [3] / [4]
class _StubDecoder:
    """--stub-decoder: the product decoder's slot-loop INTERFACE (infer_batched(..., source=, slots=, max_new_tokens=)) without a GPU:
    request i runs for max_new_tokens[i] steps in one of `slots` slots and returns that many tokens (value i mod 1024); a finished
    slot pulls the next request from the shared source -- the dealing, not the arithmetic, is what the control-path run exercises."""
    refill_ahead = 0

    def __init__(self, dev):
        self.device = dev
        self.last_stats = {"steps": 0, "kv_rows": 0, "passes": 0}

    def infer_batched(self, xs, ys, berts, source=None, slots=4, max_new_tokens=None, **kw):
        live, pred, idx, steps = {}, [], [], 0
        for s_ in range(slots):
            c = source.next()
            if c is None:
                break
            live[s_] = [c, int(max_new_tokens[c])]
        while live:
            steps += 1
            if steps % 64 == 0:
                time.sleep(0.0005 * (1 + int(os.environ.get("RANK", "0")) % 3))     # ranks of unequal speed: the cursor deals on demand
            for s_ in list(live):
                live[s_][1] -= 1
                if live[s_][1] <= 0:
                    c = live[s_][0]
                    pred.append(torch.full((int(max_new_tokens[c]),), c % 1024, dtype=torch.int64))
                    idx.append(c)
                    n = source.next()
                    if n is None:
                        del live[s_]
                    else:
                        live[s_] = [n, int(max_new_tokens[n])]
        self.last_stats = {"steps": steps, "kv_rows": 0, "passes": 0}
        return pred, torch.tensor(idx, dtype=torch.int64)


class _StubVocoder:
    samples_per_frame = 640

    def flow_dec(self, z, m, ge):
        return torch.zeros(1, 1, z.shape[2] * self.samples_per_frame)


class CBWorkload:
    """Continuous batching through the multi-GPU engine, the vocoder stage and the rank-0 gather included: what `--workload cb`
    times, and what every `--workload single` line carries as its `cb32` sub-record (v2Pro, 32 slots and 256 requests per GPU:
    BASELINE's "bs=1/32"; at --gpus 8 that sub-record IS configs[3])."""

    def __init__(self, a, world, rank, dev, dist, version, dtype_name, slots, requests, voc=None, book=None, ge=None):
        from gsv_tts_lite_amd import synth, engine
        from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
        from gsv_tts_lite_amd.sovits import _VocoderNative
        self.a, self.world, self.rank, self.dev, self.dist = a, world, rank, dev, dist
        self.version, self.dtype_name, self.slots, self.requests = version, dtype_name, slots, requests
        self.dtype = _torch_dtype(dtype_name)
        self.vdtype = torch.bfloat16 if dtype_name == "fp8" else self.dtype      # fp8 operands exist in the GPT batched step only
        self.sbytes = 4 if dtype_name == "fp32" else 2
        cfg = synth.gpt_config()
        self.hps = synth.sovits_hps(version)
        gin = self.hps["model"]["gin_channels"]
        if a.stub_decoder:
            self.t2s, voc = _StubDecoder(dev), _StubVocoder()
        else:
            gw = synth.gpt_weights(cfg, seed=1234, eos_gain=0.0)      # lengths come from the per-request budgets below, not from EOS
            self.t2s = Text2SemanticDecoder(cfg)
            self.t2s.load_state_dict(gw)
            self.t2s.initialize_runtime(self.dtype, dev, [(slots, 512), (slots, 1024)])     # SURVEY.md 8(d) buckets
        if voc is None:
            sw = synth.sovits_weights(self.hps, seed=1234, hot_path_only=True)
            voc = _VocoderNative(self.hps["model"], {k: torch.from_numpy(v) for k, v in sw.items()}, self.vdtype, dev)
        self.voc = voc
        self.eng = engine.ContinuousBatchingEngine(self.t2s, slots=slots, chunk=2)
        self.book = book if book is not None else engine.SpeakerBook(dev)
        self.ge = ge if ge is not None else self.book.sync("speaker-0", [torch.from_numpy(synth.synth_ge(0, gin, 1234))] if rank == 0 else None)[0]
        self.n_req = n_req = requests * world
        lens = synth.mixed_lengths(n_req)                         # Lx2 ~ U[20,120], Ly ~ U[75,150]   
        self.new_tok = synth.mixed_new_tokens(n_req)              # N ~ U[50,400]
        reqs = [synth.synth_request(i, 40, t, n) for i, (t, n) in enumerate(lens)]
        self.xs = [torch.from_numpy(r[0]).to(dev) for r in reqs]
        self.ys = [torch.from_numpy(r[1]).to(dev) for r in reqs]
        self.bs = [torch.from_numpy(r[2]).to(dev) for r in reqs]
        self.acc = {"tok": 0, "frames": 0, "t_ar": 0.0, "t_voc": 0.0, "steps": 0, "kv_rows": 0, "mine": 0, "gathered_samples": 0, "timed": 0}
        self.costs = [int(x.shape[0]) + int(y.shape[0]) + int(n) for x, y, n in zip(self.xs, self.ys, self.new_tok)] if a.lpt_budget else None

    def vocode(self, tokens):
        """TTS.infer_batched's vocoder stage (TTS.py:705-764, tts.py): length-balanced order over ALL requests, time-concatenated
        batches of 10 with per-frame ge, batch b on rank b mod world; -> ({request: its samples (device)}, frames vocoded here)"""
        from gsv_tts_lite_amd.batchmath import balance_order
        dev, voc, ge = self.dev, self.voc, self.ge
        lengths = torch.tensor([len(p) for p in tokens])
        order = balance_order(lengths)
        batches = [order[s:s + 10].tolist() for s in range(0, len(order), 10)]
        tot, audio = 0, {}
        for b in self.eng.deal_batches(len(batches)):
            T = int(sum(2 * int(lengths[i]) for i in batches[b]))
            if T == 0:
                for i in batches[b]:
                    audio[i] = torch.empty(0, device=dev)
                continue
            z = torch.randn(1, 192, T, device=dev)
            o = voc.flow_dec(z, torch.ones(1, 1, T, device=dev), ge.expand(-1, -1, T).contiguous())[0, 0]
            pos = 0
            for i in batches[b]:
                n_ = 2 * int(lengths[i]) * voc.samples_per_frame
                audio[i] = o[pos:pos + n_] if not self.a.stub_decoder else torch.full((n_,), float(i))   # stub: samples name their request
                pos += n_
            tot += T
        return audio, tot

    def vocode_decode(self, tokens, vq):
        """the same batches through SynthesizerTrn.decode: the requests' own tokens and target phonemes, per-token ge, slice_indices
        (quantizer lookup + device enc_p + noise + flow + Generator in one library call) -- exactly TTS.infer_batched's call"""
        from gsv_tts_lite_amd.batchmath import balance_order
        dev, xs, ge = self.dev, self.xs, self.ge
        lengths = torch.tensor([len(p) for p in tokens])
        order = balance_order(lengths)
        batches = [order[s:s + 10].tolist() for s in range(0, len(order), 10)]
        tot = 0
        for b in self.eng.deal_batches(len(batches)):
            oi = [i for i in batches[b] if int(lengths[i]) > 0]
            if not oi:
                continue
            ln = [int(lengths[i]) for i in oi]
            ph = [xs[i][40:] for i in oi]
            ends = torch.cumsum(torch.tensor([len(p) for p in ph]), 0)
            pairs = torch.stack([ends - torch.tensor([len(p) for p in ph]), ends], dim=1).to(dev)
            sl = torch.repeat_interleave(pairs, (torch.tensor(ln) * 2).to(dev), dim=0)
            vq.decode(torch.cat([tokens[i] for i in oi])[None, None], torch.cat(ph)[None], ge.expand(-1, -1, sum(ln)), noise_scale=0.5,
                      cuda_graph=False, slice_indices=sl)
            tot += 2 * sum(ln)
        return tot

    def vocode_batch(self, items):
        """one time-concatenated vocoder batch (completion order): the overlapped engine calls this on its side stream"""
        dev = self.dev
        T = int(sum(2 * len(p) for _, p in items))
        if T:
            z = torch.randn(1, 192, T, device=dev)
            self.voc.flow_dec(z, torch.ones(1, 1, T, device=dev), self.ge.expand(-1, -1, T).contiguous())
        return {i: 2 * len(p) for i, p in items}

    def step(self, i, timed_idx):
        a, dev, eng, acc, t2s = self.a, self.dev, self.eng, self.acc, self.t2s
        _sync(dev); s0 = time.perf_counter()
        if (not a.overlap):
            pred, idx = eng.run_gpt(self.xs, self.ys, self.bs, costs=self.costs, top_k=1, max_new_tokens=self.new_tok, async_refill=not a.sync_refill)
            _sync(dev); s1 = time.perf_counter()
            # every rank learns every request's tokens (ids: a few hundred KB), vocodes the batches dealt to it, and the
            # samples meet on rank 0 -- what TTS.infer_batched does between its GPT and its return (tts.py)
            tokens = eng.exchange({int(i): p for i, p in zip(idx.tolist(), pred)}, self.n_req, dst=None)
            eng._retire_cursors(None)
            audio, frames = self.vocode(tokens)
            full = eng.exchange(audio, self.n_req, dst=0)
            if a.stub_decoder and full is not None:      # the control-path run checks what the gather delivered: every request, once, in order
                for r_, t in enumerate(full):
                    assert t.numel() == 2 * int(self.new_tok[r_]) * self.voc.samples_per_frame and (t.numel() == 0 or bool((t == float(r_)).all())), r_
                acc["stub_ordered"] = acc.get("stub_ordered", 0) + len(full)
            if timed_idx is not None and full is not None:
                acc["gathered_samples"] += int(sum(t.numel() for t in full))
            del full, audio
        else:
            res, pred, idx = eng.run_overlapped(self.xs, self.ys, self.bs, self.vocode_batch, batch=10, costs=self.costs, top_k=1,
                                                max_new_tokens=self.new_tok, async_refill=not a.sync_refill)
            s1 = time.perf_counter()
            frames = int(sum(res.values()))
        _sync(dev); s2 = time.perf_counter()
        if timed_idx is not None:
            acc["tok"] += int(sum(len(p) for p in pred)); acc["frames"] += frames
            acc["t_ar"] += s1 - s0; acc["t_voc"] += s2 - s1; acc["mine"] += len(pred)
            acc["steps"] += t2s.last_stats["steps"]; acc["kv_rows"] += t2s.last_stats["kv_rows"]; acc["timed"] += 1
            # slot-steps as the loop ran them: the tail of the queue continues on smaller bound states (t2s.py `compact`)
            acc["slot_steps"] = acc.get("slot_steps", 0) + int(t2s.last_stats.get("slot_steps", t2s.last_stats["steps"] * self.slots))
            acc["compacted_steps"] = acc.get("compacted_steps", 0) + sum(1 for _ in t2s.last_stats.get("compactions", []))
            if t2s.last_stats.get("compactions"):
                acc["compactions"] = [list(c) for c in t2s.last_stats["compactions"]]
            acc["passes"] = acc.get("passes", 0) + int(t2s.last_stats.get("passes", 0))

    def refill_label(self):
        """what the slot loop that was measured does when a slot ends -- read from the decoder, not assumed"""
        a, t2s = self.a, self.t2s
        if a.sync_refill:
            return "reference order: every slot waits for the prompt pass (t2s_model.py:696-722)"
        ahead = int(getattr(t2s, "refill_ahead", 0))
        passes = self.acc.get("passes", 0) / max(1, self.acc["timed"])
        if ahead <= 0:
            return ("staged (GSV_REFILL_AHEAD=0): the prompt pass of a finished slot runs on a side stream, the slot joins at the next window "
                    "after it; %.0f packed prompt passes per step on this rank" % passes)
        sh = getattr(t2s, "_ahead", None)
        mem = 0 if sh is None else sum(v.numel() * v.element_size() for v in sh.values() if torch.is_tensor(v))
        return ("ahead: up to %d of the next requests are prefilled on a side stream into a second bound state (%d slots, %.0f MB of its own K/V "
                "cache and state) and adopted by the slot that ends (gsv_t2s_adopt_slots); %.0f packed prompt passes per step on this rank"
                % (ahead, 0 if sh is None else int(sh["slots"]), mem / 1e6, passes))

    def record(self, elapsed, steps, warmup):
        """every rank calls it (one all-reduce of the totals); -> the record (value = whole-job tokens/s over all ranks)"""
        a, acc, world, dev, t2s = self.a, self.acc, self.world, self.dev, self.t2s
        tot = torch.tensor([acc["tok"], acc["frames"], acc["mine"], acc.get("slot_steps", acc["steps"] * self.slots)], dtype=torch.float64, device=dev)
        if self.dist is not None:
            self.dist.all_reduce(tot)
        tok_all, frames_all, slot_steps_all = float(tot[0]), float(tot[1]), float(tot[3])
        assert int(tot[2]) == self.n_req * steps, "every request must have been served exactly once per step"
        value = tok_all / elapsed
        which = "configs[4]" if self.dtype_name == "fp8" else ("configs[2]" if world == 1 and self.version == "v2ProPlus" else "configs[3]")
        out = {
            "metric": "semantic_tokens_per_sec_end_to_end (GPT AR incl. prefill + flow/Generator vocoder); RTF^-1 = value/25",
            "value": value, "unit": "semantic_tokens/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": self.dtype_name, "data": "synthetic (seeded random weights of the real architecture, synthetic phoneme/token ids)",
            "config": {"workload": "%s: %s continuous batching, %d mixed-length requests per GPU per step through %d slots per GPU "
                                   "(greedy, 50..400 new tokens per request), flow/Generator over every utterance in time-concatenated batches of 10"
                                   % (which, self.version, self.requests, self.slots),
                       "requests_per_step": self.n_req, "gpt_cache": [(self.slots, 512), (self.slots, 1024)],
                       "refill": self.refill_label(),
                       "queue_order": "longest first by prompt rows + token budget" if a.lpt_budget else "longest first by text length (the budgets are independent of it)",
                       "vocoder": "after the slot loop, length-balanced batches (TTS.py:705-764)" if (not a.overlap) else
                                  "overlapped with the slot loop on a side stream, batches of 10 in completion order",
                       "parallelism": "one engine per GPU x%d, requests pulled on demand from a shared cursor, ge broadcast once per "
                                      "speaker (%d tensor broadcasts in this run)" % (world, self.book.broadcasts)},
            "dist": dist_info(a, world, self.book),
            "gather": None if a.overlap else "inside the timed step: token ids all-gathered (device), every request's samples sent to rank 0 "
                                             "(device, point-to-point); %d samples arrived on rank 0 over the %d timed steps" % (acc["gathered_samples"], steps),
            "audio_s_per_s_end_to_end": frames_all / 50.0 / elapsed,
            "tokens_per_step": tok_all / steps, "mean_tokens_per_request": tok_all / steps / self.n_req,
            "idle_slot_steps_frac": 1.0 - tok_all / max(slot_steps_all, 1.0),
            "idle_slot_steps_frac_without_tail_compaction": 1.0 - acc["tok"] / max(acc["steps"] * self.slots, 1.0),
            "tail_compaction": {"levels": list(getattr(t2s, "tail_levels", [])),
                                "moves_of_the_last_step_on_rank0 (window, from slots, to slots, live requests)": acc.get("compactions", []),
                                "note": "queue empty and nothing prefilled ahead: the live requests continue on a smaller bound state "
                                        "(gsv_t2s_move_slots); the reference keeps stepping the full batch (t2s_model.py:684-694)"},
            "rank0_ar_tokens_per_s": acc["tok"] / acc["t_ar"],
            "rank0_vocoder_audio_s_per_s": acc["frames"] / 50.0 / max(acc["t_voc"], 1e-9) if (not a.overlap) else None,
            "rank0_requests_served_per_step": acc["mine"] / steps,
        }
        if a.stub_decoder:
            out["data"] = "STUB: no kernels ran (--stub-decoder: CPU ranks, stub slot loop and vocoder); this line proves the control path, it measures nothing"
            out["stub_requests_gathered_in_order_on_rank0"] = acc.get("stub_ordered", 0)
            return out
        if self.rank == 0:
            # step-level roofline of the batched decode step on this rank: weights once per step + the K/V rows read
            wbytes = GPT_PARAMS * 2
            if self.dtype_name == "fp32":
                wbytes = GPT_PARAMS * 4
            if self.dtype_name == "fp8":   # QKV / W1 / W2 as e4m3 (+ fp32 scales); out-proj, predict layer bf16
                wbytes = (24 * (3 * 512 * 512 + 2 * 2048 * 512) * 1 + 24 * 512 * 512 * 2 + 1025 * 512 * 2 + 0.16e6 * 4 + 24 * 4096 * 4)
            by = acc["steps"] * wbytes + acc["kv_rows"] * KV_BYTES_PER_POS * self.sbytes
            gbs = by / acc["t_ar"] / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                               "traffic": _step_traffic(self.slots, self.dtype_name),
                               "kernel": ("batched decode step (5 launches per layer: GEMMs on 16 x 16 x 32 MFMA tiles + attention per (head, sequence), csrc/t2s_small.h)" if self.slots >= t2s.batched_min else
                                          "decode step, 2 launches per layer with 2 / 4 sequences per block (csrc/t2s_decode_multi.h)" if self.slots > 16
                                          else "decode step, 2 launches per layer (csrc/t2s_decode.h)"),
                               "ms_per_step_of_the_slot_loop": acc["t_ar"] / max(1, acc["steps"]) * 1e3,
                               "note": "algorithmic bytes of the AR phase = decode steps x weight bytes + K/V rows read x row bytes (prefills and "
                                       "refills are inside the time, not in the bytes) / AR wall time of rank 0"}
            vbytes, vflops = vocoder_algorithmic(self.version, 4 if self.dtype_name == "fp32" else 2)
            if acc["frames"] and (not a.overlap):
                g = vbytes * acc["frames"] / acc["t_voc"] / 1e9
                out["roofline_vocoder"] = {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                                           "traffic": None, "mfma_tflops": vflops * acc["frames"] / acc["t_voc"] / 1e12}
        return out


def run_cb(a):
    world, rank, dev, dist = setup_dist(a)
    from gsv_tts_lite_amd import synth
    w = CBWorkload(a, world, rank, dev, dist, a.version, a.dtype, a.slots, a.requests)
    log("models ready")
    elapsed = timed_region(a.steps, a.warmup, w.step, dev, dist)
    log("timed region done: %.3f s" % elapsed)
    out = w.record(elapsed, a.steps, a.warmup)
    value = out["value"]
    t2s, eng, xs, ys, bs, n_req = w.t2s, w.eng, w.xs, w.ys, w.bs, w.n_req
    if (not a.no_extras) and (not a.overlap):
        # one more pass over the queue with the vocoder stage as TTS.infer_batched runs it: decode() on the requests' own tokens
        # (every rank takes part: the token exchange is a collective)
        try:
            from gsv_tts_lite_amd.sovits import SynthesizerTrn
            vq = SynthesizerTrn(1025, 32, n_speakers=300, **w.hps["model"])
            vq.load_state_dict(synth.sovits_weights(w.hps, seed=1234))
            vq.initialize_runtime(w.vdtype, dev, [])
            # untimed: one batch through decode() (workspace allocation, kernel attributes, code load)
            wl = [50 + 17 * i for i in range(10)]
            vq.decode(torch.zeros(1, 1, sum(wl), dtype=torch.int64, device=dev), torch.cat([xs[i][40:] for i in range(10)])[None],
                      w.ge.expand(-1, -1, sum(wl)), noise_scale=0.5, cuda_graph=False)
            _sync(dev)
            if dist is not None:
                dist.barrier()
            q0 = time.perf_counter()
            pred, idx = eng.run_gpt(xs, ys, bs, costs=w.costs, top_k=1, max_new_tokens=w.new_tok, async_refill=not a.sync_refill)
            tokens = eng.exchange({int(i): p for i, p in zip(idx.tolist(), pred)}, n_req, dst=None)
            eng._retire_cursors(None)
            w.vocode_decode(tokens, vq)
            _sync(dev)
            from gsv_tts_lite_amd import scheduler
            dt = scheduler.max_over_ranks(time.perf_counter() - q0, device=dev)
            out["value_with_enc_p"] = float(sum(len(p) for p in tokens)) / dt
            out["value_with_enc_p_note"] = ("one pass over the queue with SynthesizerTrn.decode (enc_p + noise + flow + Generator, per-token ge, "
                                            "slice_indices) as the vocoder stage instead of flow_dec on a synthetic z_p; no audio gather in it")
            del vq
        except Exception as e:  # noqa: BLE001
            log("value_with_enc_p pass failed: %r" % (e,))
    if rank == 0 and not a.no_extras:
        # BASELINE's "p50 TTFT ... bs=32": the first `slots` requests of the queue arrive together -> packed prompt pass of
        # all of them + the first decode step (every request's first token exists), outside the timed region
        try:
            first = list(range(min(a.slots, n_req)))
            ts = []
            for _ in range(12):
                _sync(dev); q0 = time.perf_counter()
                with torch.inference_mode():
                    xy, xl, yl, _, _ = t2s.embed_prompt([xs[c] for c in first], [ys[c] for c in first], [bs[c] for c in first])
                    t2s.prefill(a.slots, 0, xy, xl, yl)
                    t2s._decode(a.slots, 1)
                _sync(dev); ts.append(time.perf_counter() - q0)
            out["ttft_ms_p50_first_batch"] = sorted(ts[2:])[len(ts[2:]) // 2] * 1e3
            out["ttft_first_batch_note"] = "%d prompts (%d positions in all) in one packed prompt pass + the first decode step" % (
                len(first), int(sum(int(xs[c].shape[0]) + int(ys[c].shape[0]) for c in first)))
        except Exception as e:  # noqa: BLE001
            out["ttft_ms_p50_first_batch"] = None
            log("first-batch TTFT failed: %r" % (e,))
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (subprocess) ...")
            cb = cpu_baseline("cb", a.version)
            out["cpu_baseline"] = cb
            if cb.get("value"):
                out["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.cpu_baseline_worker:
        cpu_baseline_worker(a.cpu_baseline_worker, a.version)
        return
    if a.stub_decoder:
        a.no_extras = a.no_cpu_baseline = True
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    if a.workload == "cb":
        if a.steps == 10 and a.warmup == 2 and "--steps" not in sys.argv:
            a.steps, a.warmup = 3, 1      # a step is a whole queue (~2 s): keep the default run within minutes
        run_cb(a)
    else:
        run_single(a)


if __name__ == "__main__":
    main()
