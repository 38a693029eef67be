#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / avg / min / max / share."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by 6 desc"))
tot = sum(r[5] for r in rows)
print("%-100s %8s %10s %9s %10s %6s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "%"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-100s %8d %10.1f %9d %10d %6.2f" % (r[0][:100], r[1], r[2], r[3], r[4], 100 * r[5] / tot))
