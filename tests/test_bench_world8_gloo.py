"""`bench.py --gpus 8 --workload cb` without a GPU: the control path of the 8-GPU run (bench.py's own launcher, one process per rank,
gloo on 127.0.0.1) with a stub slot loop and a stub vocoder (--stub-decoder).  What runs for real: the launcher and its rank
watching, the process group, the shared request cursor (engine.RequestSource over the store), the token exchange, the vocoder
batches dealt over the ranks, the point-to-point gather of every request's samples on rank 0, `max_over_ranks`, and the record's
all-reduce + "every request exactly once" assertion.  The N > 1 hardware run is the driver's; this is what can be proven here."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=420):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "cb", "--dist-backend", "gloo", "--stub-decoder",
           "--steps", "1", "--warmup", "1", "--requests", "6", "--slots", "3"] + extra
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    t0 = time.perf_counter()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    return p, time.perf_counter() - t0


def test_world8_cb_control_path_every_request_once_and_in_order():
    p, _ = _run([])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE line, the other ranks none: %r" % (p.stdout[-2000:],)
    rec = json.loads(lines[0])
    sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
    from gsv_tts_lite_amd import synth
    n_req = 8 * 6
    new_tok = synth.mixed_new_tokens(n_req)
    assert rec["n_gpus"] == 8 and rec["dist"]["world_size"] == 8 and rec["dist"]["backend"] == "gloo"
    assert rec["data"].startswith("STUB"), "a control-path line must say that it measured nothing"
    assert rec["config"]["requests_per_step"] == n_req
    assert rec["tokens_per_step"] == float(np.sum(new_tok)), "every request's tokens, once"
    # the rank-0 gather delivered every request of the warm-up pass and of the timed pass, each checked sample for sample
    assert rec["stub_requests_gathered_in_order_on_rank0"] == 2 * n_req
    assert 0 < rec["rank0_requests_served_per_step"] < n_req, "eight ranks pulled from ONE cursor"
    assert str(8 * 0 + int(np.sum(new_tok)) * 2 * 640) in rec["gather"], rec["gather"]


def test_a_rank_that_dies_stops_the_run_with_a_message():
    p, dt = _run(["--stub-die-rank", "5"], timeout=180)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert "rank 5 exited with code 3" in p.stderr and "were stopped" in p.stderr
    assert dt < 120, "the launcher must not wait for a collective's timeout (%.0f s)" % dt
