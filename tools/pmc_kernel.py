"""Mean PMC counter values per kernel from rocprofv3 --pmc passes (CSV output), any command.
usage: python tools/pmc_kernel.py <kernel substring> <dir> [<dir> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

key = sys.argv[1]
for d in sys.argv[2:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if key in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.path.basename(d) + ": " + "  ".join(f"{c} {sum(v[-4:]) / len(v[-4:]):.6g} (n={len(v)})" for c, v in sorted(agg.items())))
