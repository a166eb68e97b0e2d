"""Timeline of one training step from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv).
usage: python tools/trace_csv.py <kernel_trace.csv> [step] [--sum]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kind"] == "KERNEL_DISPATCH"]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
step = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8
# a step ends with the optimizer update (the voxeliser is no step boundary any more: PipelinedStep runs the NEXT sweep's
# voxelisation on the second stream in the middle of the backward pass)
# (two optimizer launches per step since round 4: the RPN + head variables early, the rest -- the SMALLER grid -- at the end)
sgd = [r for r in rows if "k_sgd_nesterov" in r["Kernel_Name"]]
gmin = min(int(r["Grid_Size_X"]) for r in sgd)
idx = [i + 1 for i, r in enumerate(rows) if "k_sgd_nesterov" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == gmin]
a, b = idx[step], idx[step + 1]
t0 = rows[a - 1]["e"]
agg = defaultdict(lambda: [0, 0.0])
busy, cur_s, cur_e = 0.0, None, None
prev_end = {}
for r in rows[a:b]:
    n = re.sub(r"\(anonymous namespace\)::|lisec::|void ", "", r["Kernel_Name"]).split("(")[0]
    d = (r["e"] - r["s"]) / 1000.0
    agg[n][0] += 1
    agg[n][1] += d
    if cur_e is None or r["s"] > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = r["s"], r["e"]
    else:
        cur_e = max(cur_e, r["e"])
    if "--sum" not in sys.argv:
        q = r["Queue_Id"]
        gap = (r["s"] - prev_end.get(q, r["s"])) / 1000.0
        prev_end[q] = r["e"]
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        nwg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
        print("%7.0f %7.1f gap %5.1f q%s %-34s wgs %6d" % ((r["s"] - t0) / 1000.0, d, gap, q, n[:34], nwg))
busy += cur_e - cur_s
print("step span %.0f us (end of one optimizer update to the end of the next), busy (union) %.0f us, kernel sum %.0f us" % ((rows[b - 1]["e"] - t0) / 1000.0, busy / 1000.0,
                                                                   sum(v[1] for v in agg.values())))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-36s %4d %8.1f" % (n[:36], c, t))
