#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU (CSV) -> per kernel, per dispatch:
where the wave-cycles of the latency-bound decode kernels go (MI355X_MICROARCH.md: WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue
stall, ACTIVE_INST_ANY = issuing; the three sum to ~WAVE_CYCLES) and how many vector instructions a wave issues.
usage: tools/pmc_valu.py <counter_collection.csv>"""
import csv, re, sys
from collections import defaultdict
def short(n):
    m = re.search(r"gsv::(\w+)(<[^>]*>)?", n)
    return ((m.group(1) + (m.group(2) or "")) if m else n.strip()[:40])[:60]
acc = defaultdict(lambda: defaultdict(list)); waves = {}
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        try: waves[k] = int(row["Grid_Size"]) // 64
        except Exception: pass
print("%-62s %8s %12s %8s %8s %8s %10s %12s" % ("kernel", "launches", "wave_cycles", "parked", "stalled", "issuing", "valu_busy", "valu/wave"))
rows = []
for k, c in acc.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0: continue
    rows.append((wc, k, len(next(iter(c.values()))), a))
for wc, k, n, a in sorted(rows, key=lambda r: -r[0])[:24]:
    w = waves.get(k, 0)
    print("%-62s %8d %12.0f %7.1f%% %7.1f%% %7.1f%% %9.1f%% %12s" % (k, n, wc, 100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc,
          100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * a.get("SQ_ACTIVE_INST_VALU", 0) / wc, "%.0f" % (a.get("SQ_INSTS_VALU", 0) / w) if w else "-"))
