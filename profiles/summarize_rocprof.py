#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats and PMC counter_collection) for the mrhip kernels.
usage: summarize_rocprof.py <dir> [<dir> ...]   -> prints compact JSON (kernel names shortened)"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    m = re.search(r"(mrhip::\(anonymous namespace\)::)?(\w+_kernel<[^>]*>|\w+_kernel)", name)
    return m.group(2) if m and "mrhip" in name else ("torch:" + name[:40] if "at::" in name else name[:60])

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        out.setdefault(d, {})["kernel_stats"] = [
            {"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
             "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2),
             "pct": float(r["Percentage"])} for r in rows]
    # The stats file averages over EVERY launch of the run, warm-up included (the first launches of a process run at lower
    # clocks); the bench's HIP events bracket only the timed region = the LAST launches.  From the kernel trace, in launch
    # order: the average of the last 5 launches of each mrhip kernel (bench.py's default --steps) next to the run's average.
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        runs = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mrhip" in r["Kernel_Name"]:
                runs[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        tl = {}
        for k, v in runs.items():
            v.sort()
            dur = [(e - b) / 1e3 for b, e in v]
            tl[k] = {"launches": len(dur), "all_avg_us": round(sum(dur) / len(dur), 2), "last5_avg_us": round(sum(dur[-5:]) / len(dur[-5:]), 2),
                     "first3_us": [round(x, 1) for x in dur[:3]]}
        out.setdefault(d, {})["kernel_trace"] = tl
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith(("poly", "arb", "shiftin", "deci", "rational", "fir", "interp", "farrow", "sched")):
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            # (rocprofv3's VGPR_Count / LDS_Block_Size columns are NOT reported here: for these kernels they read 40 VGPRs
            #  and 0 LDS while the code object says 76-131 VGPRs and the launch uses tens of KB of dynamic LDS --
            #  occupancy must be taken from the code-object metadata and the launch's own debug line, MRHIP_DEBUG=1)
            meta[k] = {"wg": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
        out.setdefault(d, {})["counters_avg_per_dispatch"] = {
            k: dict(meta[k], dispatches=len(next(iter(v.values()))), **{c: round(sum(x) / len(x), 1) for c, x in v.items()})
            for k, v in agg.items()}
print(json.dumps(out, indent=1))
