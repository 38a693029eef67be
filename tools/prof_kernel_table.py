#!/usr/bin/env python3
"""per-kernel totals of a rocprofv3 rocpd db over the LAST `frac` of the trace (steady state)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = list(db.execute("select name,start,end,grid_x,workgroup_x from kernels order by start"))
t0 = rows[0][1]; t1 = rows[-1][2]; cut = t1 - (t1 - t0) * frac
agg = {}
for n, s, e, gx, wx in rows:
    if s < cut: continue
    short = n.split('gsv::')[1][:50] if 'gsv::' in n else n[:50]
    a = agg.setdefault(short, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print("window %.2f ms, kernel time %.2f ms" % ((t1 - cut) / 1e6, tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-52s x%5d  total %9.1f us  avg %7.1f us  %5.1f%%" % (k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
