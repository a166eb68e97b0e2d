"""Summarise rocprofv3 --pmc passes of `bench.py` (one counter set per pass, CSV output): per kernel of interest the
mean counter value and duration over its last launches.
usage: python tools/pmc_summary.py <dir> [<dir> ...]
       python tools/pmc_summary.py --dominant profiles/r03_pmc_dominant.json <dir> [<dir> ...]
The second form also writes what bench.py quotes in its `roofline` object for the dominant kernel (the mid2 forward in the
Winograd form on its own symbol): traffic_bytes = FETCH_SIZE x 2 (the gfx950 correction for wide coalesced reads, MI355X guide) +
WRITE_SIZE, in bytes per launch; clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /
(clock x duration x 1024 SIMDs)."""
import csv
import glob
import os
import sys
from collections import defaultdict

DOMINANT = "mid2 Conv3D forward, Winograd form (roofline launch)"
KEYS = {"k_wino<false, 0, 1>": DOMINANT, "k_igemm_halo<1, false, 1, 2": "mid2 Conv3D data gradient, direct form (roofline launch)",
        "k_wino<false, 0, 0>": "Winograd contractions without on-load BN (mid blocks, data gradients)", "k_wino<true, 0, 0>": "Winograd contractions with on-load BN (rpn1 forward)",
        "k_wino_wgrad(": "Winograd weight gradient (mid2, mid3)", "k_wino_wgrad_sum": "Winograd weight gradient: slab sum", "k_wgrad_halo<false, 7>": "mid wgrad (halo)", "k_field_taps": "field conv: taps",
        "k_field_combine": "field conv: combine", "k_igemm_halo<0, false, 1, 2>": "mid1 Conv3D fwd (roofline launch)", "k_vfe_grid": "VFE grid writer",
        "k_vfe_stage<2": "VFE layers 1+2", "k_vfe_stage<3": "VFE layer 3", "k_wgrad_halo<false>": "mid wgrad (halo)",
        "k_wgrad_ring<false, 5>": "mid wgrad (ring)", "k_wgrad_ring_batch<true, 5>": "rpn1/rpn2 batched wgrad (ring)",
        "k_wgrad_ring_batch<true, 3>": "rpn3 batched wgrad (ring)", "k_wgrad_ring_reduce": "ring slab sum"}
args = sys.argv[1:]
dominant_out = None
if args and args[0] == "--dominant":
    dominant_out, args = args[1], args[2:]
dom = {}
vfe = {}
for d in args:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
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
            if label == DOMINANT:
                for c, v in cs.items():
                    tail = v[-5:]
                    dom[c] = sum(tail) / len(tail)
                    if c == "__dur_us":
                        dom.setdefault("__dur_by_pass", {})[os.path.basename(d)] = dom[c]
            if label == "VFE grid writer":
                # the launches WITH the dense grid are the long ones (the step's launches write per-voxel outputs only)
                for c, v in cs.items():
                    if c in ("FETCH_SIZE", "WRITE_SIZE"):
                        big = sorted(v)[-max(1, len(v) // 4):]
                        vfe[c] = sum(big) / len(big)
if dominant_out:
    import json
    out = {"kernel": "k_wino<false,0,1> (mid2 Conv3D forward in the Winograd form, bench.py's roofline launch)",
           "source": "rocprofv3 --pmc passes of `python bench.py --steps 3 --warmup 2 --no-cpu-baseline`, one counter set per pass "
                     "(tools/profile_round.sh), summarised by tools/pmc_summary.py"}
    if "FETCH_SIZE" in dom and "WRITE_SIZE" in dom:
        out["fetch_size_kib"], out["write_size_kib"] = dom["FETCH_SIZE"], dom["WRITE_SIZE"]
        out["traffic_bytes"] = (2.0 * dom["FETCH_SIZE"] + dom["WRITE_SIZE"]) * 1024.0
    if "GRBM_GUI_ACTIVE" in dom and "__dur_us" in dom:
        dur = dom.get("__dur_by_pass", {}).get("pmc_clk", dom["__dur_us"])
        out["clock_ghz"] = dom["GRBM_GUI_ACTIVE"] / 8.0 / dur / 1e3
        out["us_per_launch_under_counters"] = dur
        if "SQ_VALU_MFMA_BUSY_CYCLES" in dom:
            dur_sq = dom.get("__dur_by_pass", {}).get("pmc_sq", dur)
            out["mfma_busy"] = dom["SQ_VALU_MFMA_BUSY_CYCLES"] / (out["clock_ghz"] * 1e3 * dur_sq * 1024.0)
            out["mfma_instructions"] = dom.get("SQ_INSTS_MFMA")
    if "FETCH_SIZE" in vfe and "WRITE_SIZE" in vfe:
        out["vfe_grid"] = {"fetch_size_kib": vfe["FETCH_SIZE"], "write_size_kib": vfe["WRITE_SIZE"],
                           "traffic_bytes": (2.0 * vfe["FETCH_SIZE"] + vfe["WRITE_SIZE"]) * 1024.0,
                           "note": "k_vfe_grid launches with the dense (8,200,400,64) grid (the top quarter by bytes)"}
    tl = os.environ.get("STEP_TIMELINE")
    if tl and os.path.exists(tl):
        # the dominant layer inside the step: the longest 1250-workgroup launch of the mode-1 two-line halo kernel
        best = None
        for ln in open(tl):
            f = ln.split()
            if best is None and "k_wino<false, 0, 0>" in ln and ln.rstrip().endswith("650"):   # mid2 forward: the step's first 650-block launch
                best = float(f[1])
        if best:
            out["in_step_us"] = best
    json.dump(out, open(dominant_out, "w"), indent=1)
    print("wrote", dominant_out, out)
