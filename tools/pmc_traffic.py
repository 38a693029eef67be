#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV) into per-kernel HBM traffic per launch.

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies the 128-B requests of a wide coalesced stream at 64 B, i.e. reports exactly 1/2 of the
bytes fetched -> doubled here; WRITE_SIZE is taken as is (uncalibrated).  Infinity-Cache hits are
counted, so for weight sets that fit the 256 MiB cache this is fabric traffic, not DRAM traffic.
usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [<step_fetch.csv> <step_write.csv>]
The optional second pair is a PMC pass of `tools/step_time.py 32 bf16` (the batched decode step of the cb32 record): every kernel's
traffic summed and divided by the steps run (one t2s_token_kernel launch per step, prompt-pass kernels excluded) -> "batched_step_b32"."""
import csv, json, re, sys
from collections import defaultdict

def per_kernel(path):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc

def short(name):
    m = re.search(r"gsv::(\w+)", name)
    return m.group(1) if m else name[:40]

def step_entry(fetch_csv, write_csv):
    """(bytes per batched decode step, meta) from a PMC pass of tools/step_time.py 32 bf16"""
    sf, sw = per_kernel(fetch_csv), per_kernel(write_csv)
    # the decode steps only: the chain's kernels, the token / logits / final-LN kernels; the prompt pass (bgemm / prefill / embed) is set-up
    def _is_step(full):
        return any(t in full for t in ("sgemm_", "t2s_batch_attn", "t2s_token_kernel", "t2s_logits_kernel", "ln_rows_kernel"))
    steps = sum(len(v) for k, v in sf.items() if "t2s_token_kernel" in k)
    if not steps:
        return None, None
    tot = sum(2.0 * 1024.0 * sum(v) for k, v in sf.items() if _is_step(k)) + sum(1024.0 * sum(v) for k, v in sw.items() if _is_step(k))
    per = {short(k): (2.0 * 1024.0 * sum(v) / len(v) + 1024.0 * sum(sw.get(k, [0.0])) / max(1, len(sw.get(k, [0.0])))) for k, v in sf.items() if _is_step(k)}
    return tot / steps, {"steps_counted": steps, "bytes_per_launch": per,
                         "command": "GSV_NO_GRAPH=1 GSV_STEPS=20 GSV_PROMPT_TOK=250 tools/step_time.py 32 bf16 under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE "
                                    "(kv 350-375; eager launches: the FETCH_SIZE pass of the graph-replayed chain hangs under the profiler)"}

if sys.argv[1] == "--add-step":      # tools/pmc_traffic.py --add-step <traffic.json> <step_fetch.csv> <step_write.csv>
    tr = json.load(open(sys.argv[2]))
    val, m = step_entry(sys.argv[3], sys.argv[4])
    if val is None:
        sys.exit("no decode step found in the counter files")
    tr["batched_step_b32"] = val
    tr["_meta"]["batched_step_b32"] = m
    json.dump(tr, open(sys.argv[2], "w"), indent=1)
    print("batched decode step at 32 sequences: %.1f MB per step over %d steps" % (val / 1e6, m["steps_counted"]))
    for k, v in sorted(m["bytes_per_launch"].items(), key=lambda kv: -kv[1]):
        print("  %-28s %10.0f B per launch" % (k, v))
    sys.exit(0)
fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {}
for k, v in fetch.items():
    w = write.get(k, [0.0])
    s = short(k)
    f_bytes = 2.0 * 1024.0 * sum(v) / len(v)
    w_bytes = 1024.0 * sum(w) / len(w)
    e = out.setdefault(s, {"launches": 0, "fetch_bytes_per_launch": 0.0, "write_bytes_per_launch": 0.0})
    n0, n1 = e["launches"], len(v)
    e["fetch_bytes_per_launch"] = (e["fetch_bytes_per_launch"] * n0 + f_bytes * n1) / (n0 + n1)
    e["write_bytes_per_launch"] = (e["write_bytes_per_launch"] * n0 + w_bytes * n1) / (n0 + n1)
    e["launches"] = n0 + n1
flat = {k: v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"] for k, v in out.items()}
# whole flow + Generator pass: every launch of the vocoder-only kernel classes, divided by the number of passes
# (one conv_post / one cl->cf per pass)
def _is_voc(full):
    return ("wconv_kernel" in full or "wdma_kernel" in full or "cgemm_kernel" in full or "rbfuse_kernel" in full or "wups_kernel" in full
            or "flowfuse_kernel" in full or "flowstage_kernel" in full or "flowmerge_" in full or "avg3_kernel" in full or "conv_post_kernel" in full
            or "cf_to_cl_kernel" in full or ("tapgemm_kernel<unsigned short, unsigned short" in full)
            or "rowgemm_kernel<unsigned short, float, 16>" in full)   # cond GEMV (gin 1024); <.., 8> is also the prefill's W2
passes = sum(len(v) for k, v in fetch.items() if "conv_post_kernel" in k)
if passes:
    tot = sum(2.0 * 1024.0 * sum(v) for k, v in fetch.items() if _is_voc(k)) + sum(1024.0 * sum(v) for k, v in write.items() if _is_voc(k))
    flat["vocoder_pass"] = tot / passes
    flat["vocoder_passes_counted"] = passes
step_meta = None
if len(sys.argv) > 5:
    val, step_meta = step_entry(sys.argv[4], sys.argv[5])
    if val is not None:
        flat["batched_step_b32"] = val
import hashlib, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
d = os.path.join(root, "gsv-tts-lite_amd", "csrc")
for f in sorted(os.listdir(d)):
    if f.endswith((".h", ".hip")):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
stamp = os.path.join(root, ".commit_stamp")      # written by tools/gpu.sh before the snapshot leaves (the GPU box has no .git)
meta = {"commit": open(stamp).read().strip() if os.path.exists(stamp) else None, "csrc_sha16": h.hexdigest()[:16],
        "command": "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-cb32 under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE"}
if step_meta:
    meta["batched_step_b32"] = step_meta
json.dump({"_meta": meta, "_detail": out, **flat}, open(sys.argv[3], "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k]["launches"])[:12]:
    print("%-28s launches %6d  fetch %10.0f B  write %9.0f B" % (k, out[k]["launches"], out[k]["fetch_bytes_per_launch"], out[k]["write_bytes_per_launch"]))
