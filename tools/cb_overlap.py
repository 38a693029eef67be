#!/usr/bin/env python3
"""What the slot loop's steps lose to the refills' prompt passes: from a rocprofv3 --kernel-trace rocpd db of
`bench.py --workload cb`, the span of every decode step (logits kernel to logits kernel), split into steps a prompt-pass
kernel overlapped and steps none did; the prompt passes' own kernel time; gaps between a step's dependent launches.

    python tools/cb_overlap.py <db>
"""
import sqlite3, sys, bisect
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
DEC = ("sgemm_", "t2s_batch_attn", "t2s_logits", "t2s_token", "ln_rows", "t2s_attn_kernel", "t2s_ffn")
PRE = ("bgemm", "prefill_attn", "t2s_embed", "tapgemm", "commit", "t2s_prefill")
dec = [r for r in rows if any(k in r[0] for k in DEC)]
pre = [r for r in rows if any(k in r[0] for k in PRE) and not any(k in r[0] for k in DEC)]
lg = [r for r in dec if "t2s_logits" in r[0]]
print("%d kernels, %d decode-step kernels, %d prompt-pass kernels, %d steps" % (len(rows), len(dec), len(pre), len(lg)))
pre_s = [r[1] for r in pre]
def overlapped(a, b):      # any prompt-pass kernel intersecting [a, b)
    i = bisect.bisect_left(pre_s, a)
    if i < len(pre) and pre[i][1] < b: return True
    j = i - 1
    while j >= 0 and j >= i - 8:
        if pre[j][2] > a: return True
        j -= 1
    return False
clean, dirty = [], []
for p, q in zip(lg, lg[1:]):
    span = (q[2] - p[2]) / 1e3
    if span > 3000: continue          # a pass boundary (vocoder, warm-up) between them
    (dirty if overlapped(p[2], q[2]) else clean).append(span)
def stat(v):
    v = sorted(v); n = len(v)
    return "n %6d  mean %7.1f us  p50 %7.1f  p90 %7.1f  sum %8.1f ms" % (n, sum(v) / max(n, 1), v[n // 2] if n else 0, v[int(n * .9)] if n else 0, sum(v) / 1e3)
print("steps no prompt-pass kernel overlapped:", stat(clean))
print("steps a prompt-pass kernel overlapped :", stat(dirty))
if clean and dirty:
    m = sorted(clean)[len(clean) // 2]
    print("time the overlapped steps took beyond the clean median: %.1f ms; prompt-pass kernel time %.1f ms in %d kernels (%d commits = passes)" % (
        sum(d - m for d in dirty) / 1e3, sum(r[2] - r[1] for r in pre) / 1e6, len(pre), sum(1 for r in pre if "commit" in r[0])))
# per decode kernel class: average duration in clean vs overlapped steps
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0, 0, 0.0])
for r in dec:
    k = r[0].split("gsv::")[-1][:40]
    o = overlapped(r[1], r[2])
    a = acc[k]; a[2 * o] += 1; a[2 * o + 1] += (r[2] - r[1]) / 1e3
for k, a in sorted(acc.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:8]:
    print("%-42s alone x%6d avg %6.2f us | beside a prompt-pass kernel x%6d avg %6.2f us" % (k, a[0], a[1] / max(a[0], 1), a[2], a[3] / max(a[2], 1)))
# gaps between consecutive decode kernels inside clean steps (host starvation / window boundaries)
gaps = [(b[1] - a[2]) / 1e3 for a, b in zip(dec, dec[1:]) if 0 < b[1] - a[2] < 3e6]
big = [g for g in gaps if g > 10]
print("gaps between consecutive decode kernels: mean %.2f us; %d gaps > 10 us, sum %.1f ms (of %.1f ms decode span)" % (
    sum(gaps) / max(len(gaps), 1), len(big), sum(big) / 1e3, (dec[-1][2] - dec[0][1]) / 1e6))
