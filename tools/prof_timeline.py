#!/usr/bin/env python3
"""Print the kernel timeline of the last flow_dec (or prefill) call from a rocprofv3 rocpd db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); what = sys.argv[2] if len(sys.argv) > 2 else "voc"
rows = list(db.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x from kernels order by start"))
if what == "vocpass":   # tools/voc_time.py: passes back to back, each starts with two cf_to_cl launches; take the 10th (T = 500)
    starts = [i for i, r in enumerate(rows) if 'cf_to_cl' in r[0]][0::2]
    rows = rows[:starts[11]]
    i0 = starts[10]; stop = lambda n: False
elif what == "voc":
    idx = [i for i, r in enumerate(rows) if 'cf_to_cl' in r[0]]
    i0 = idx[-2]; stop = lambda n: 't2s_' in n
else:
    idx = [i for i, r in enumerate(rows) if 't2s_embed_kernel' in r[0]]
    i0 = idx[-1] - 1; stop = lambda n: 't2s_token' in n
seq = []
for r in rows[i0:]:
    if stop(r[0]): break
    seq.append(r)
print(len(seq), 'kernels, sum %.3f ms, span %.3f ms' % (sum(r[2]-r[1] for r in seq)/1e6, (seq[-1][2]-seq[0][1])/1e6))
from collections import OrderedDict
agg = OrderedDict()
for r in seq:
    n = r[0]; short = n.split('gsv::')[1][:44] if 'gsv::' in n else n[:44]
    key = (short, r[3]//r[6], r[4], r[5])
    agg.setdefault(key, []).append((r[2]-r[1])/1e3)
for k, v in agg.items():
    print("%-46s grid %6d %3d %2d  x%3d  avg %8.1f us  total %8.1f us" % (k[0], k[1], k[2], k[3], len(v), sum(v)/len(v), sum(v)))
