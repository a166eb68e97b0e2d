"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace sqlite database.
usage: python tools/trace_step.py gpurun_out/prof_x/x_results.db [step] [--sum]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 10
rows = db.execute("select name,start,end,stream_id,grid_x,grid_y,grid_z,workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'key_count' in r[0]]
a, b = idx[step], idx[step + 1]
t0 = rows[a][1]
agg = defaultdict(lambda: [0, 0.0])
busy, cur_s, cur_e = 0.0, None, None
for r in rows[a:b]:
    n = re.sub(r'\(anonymous namespace\)::|lisec::|void ', '', r[0]).split('(')[0]
    d = (r[2] - r[1]) / 1000.
    agg[n][0] += 1
    agg[n][1] += d
    if cur_e is None or r[1] > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = r[1], r[2]
    else:
        cur_e = max(cur_e, r[2])
    if '--sum' not in sys.argv:
        print('%7.0f %6.1f s%d %-28s grid %d,%d,%d' % ((r[1] - t0) / 1000., d, r[3], n[:28], r[4] // r[7], r[5], r[6]))
busy += cur_e - cur_s
print('step span %.0f us, busy (union) %.0f us, kernel sum %.0f us' % ((rows[b][1] - t0) / 1000., busy / 1000.,
                                                                   sum(v[1] for v in agg.values())))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%-32s %4d %8.1f' % (n[:32], c, t))
