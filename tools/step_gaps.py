#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace rocpd db of tools/step_time.py: the idle time between consecutive kernels of the hipGraph-replayed decode
steps, split into gaps INSIDE a step (node to node of one graph launch) and gaps BETWEEN steps (last kernel of one replay -> first of the next).
usage: tools/step_gaps.py <db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
# the timed replays are the last long run of decode kernels: take the last 60 steps' worth
names = [r[0] for r in rows]
last = [i for i, n in enumerate(names) if 't2s_logits_kernel' in n]
if len(last) < 62:
    print("too few steps in the trace"); sys.exit(1)
i0, i1 = last[-61] + 1, last[-1] + 1
seq = rows[i0:i1]
inner, outer, kern = [], [], 0
for a, b in zip(seq[:-1], seq[1:]):
    g = (b[1] - a[2]) / 1e3
    (outer if 't2s_logits_kernel' in a[0] else inner).append(g)
kern = sum(r[2] - r[1] for r in seq) / 1e3
span = (seq[-1][2] - seq[0][1]) / 1e3
import statistics as st
print("60 steps: span %.1f us per step, kernels %.1f us per step, %d launches per step" % (span / 60, kern / 60, len(seq) / 60))
print("gap inside a step:   n %5d  mean %.2f us  median %.2f  p90 %.2f" % (len(inner), st.mean(inner), st.median(inner), sorted(inner)[int(0.9 * len(inner))]))
print("gap between steps:   n %5d  mean %.2f us  median %.2f  p90 %.2f" % (len(outer), st.mean(outer), st.median(outer), sorted(outer)[int(0.9 * len(outer))]))
