#!/usr/bin/env python3
"""rocprofv3 --pmc (several counters, CSV) + an un-instrumented kernel-trace summary (tools/prof_summary.py output)
-> per-kernel MFMA utilisation and LDS bank-conflict share.
usage: tools/pmc_table.py <counter_collection.csv> <kernel_trace_stats.txt> [clock_GHz=2.4]
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (avg_duration * clock * 1024 SIMDs): busy cycles are summed over SIMDs,
32 per v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md).  Durations come from the trace WITHOUT counters: under --pmc
every dispatch is serialised and GRBM_GUI_ACTIVE includes ~70 us of profiler overhead per kernel, so it cannot be the denominator.
LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import csv, re, sys
from collections import defaultdict
def short(n):
    m = re.search(r"gsv::(\w+)(<[^>]*>)?", n)
    return ((m.group(1) + (m.group(2) or "")) if m else n.strip()[:40])[:64]
clock = float(sys.argv[3]) if len(sys.argv) > 3 else 2.4
dur = {}
for line in open(sys.argv[2]):
    m = re.match(r"(.*?)\s+(\d+)\s+([\d.]+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$", line)
    if m: dur.setdefault(short(m.group(1)), float(m.group(3)))
acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-66s %9s %12s %9s %9s" % ("kernel", "avg_us", "mfma_busy", "mfma_util", "lds_confl"))
rows = []
for k, c in acc.items():
    avg = {name: sum(v) / len(v) for name, v in c.items()}
    mf = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); lc = avg.get("SQ_LDS_BANK_CONFLICT", 0.0); la = avg.get("SQ_LDS_IDX_ACTIVE", 0.0)
    d = dur.get(k)
    rows.append((mf, k, d, 100 * mf / (d * clock * 1024) if d else float("nan"), 100 * lc / la if la else 0.0))
for mf, k, d, u, l in sorted(rows, key=lambda r: -r[0]):
    if mf == 0 and l < 5: continue
    print("%-66s %9s %12.0f %8.1f%% %8.1f%%" % (k, "%.1f" % (d / 1e3) if d else "-", mf, u, l))
