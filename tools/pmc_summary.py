"""Summarise rocprofv3 --pmc passes of `bench.py` (one counter set per pass, CSV output): per kernel of interest the
mean counter value and duration over its last launches.  usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

KEYS = {"k_igemm_halo<1, false, 1, 2": "mid2 Conv3D data gradient (roofline launch)", "k_field_taps": "field conv: taps",
        "k_field_combine": "field conv: combine", "k_igemm_halo<0, false, 1, 2>": "mid1 Conv3D fwd (roofline launch)", "k_vfe_grid": "VFE grid writer",
        "k_vfe_stage<2": "VFE layers 1+2", "k_vfe_stage<3": "VFE layer 3", "k_wgrad_halo<false>": "mid wgrad (halo)"}
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            for k, label in KEYS.items():
                if k in name:
                    agg[label][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if "Start_Timestamp" in r and r.get("End_Timestamp"):
                        agg[label]["__dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for label, cs in agg.items():
            parts = []
            for c, v in cs.items():
                tail = v[-5:]
                parts.append(f"{c} {sum(tail) / len(tail):.6g} (n={len(v)})")
            print(f"{os.path.basename(d)} | {label}: " + "  ".join(parts))
