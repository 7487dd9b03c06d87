#!/bin/bash
# Clock (GRBM_GUI_ACTIVE / 8 / duration) and VALU occupancy of the Float64 and mixed-precision instantiations on launches long
# enough for the quotient (64 ch x 1e7), with random and zero-filled data.  usage: bash scripts/exp_clk_f64.sh   (on the GPU box)
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
cd /tmp && export TMPDIR=/tmp
for z in 0 1; do
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/s2_clk64_$z -o clk -- python3 $R/scripts/exp_one.py --dtype float64 --long 10000000 --reps 4 --zeros $z > $R/gpurun_out/s2_clk64_$z.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/s2_clkmix_$z -o clk -- python3 $R/scripts/exp_one.py --dtype float32 --taps64 1 --long 10000000 --reps 4 --zeros $z > $R/gpurun_out/s2_clkmix_$z.log 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, os, re, collections
csv.field_size_limit(1 << 30)
for d in sorted(glob.glob("gpurun_out/s2_clk*_[01]")):
    rows = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "mrhip" not in r["Kernel_Name"] or "opair" not in r["Kernel_Name"]: continue
            key = r["Dispatch_Id"]
            rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["_dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    vs = [v for v in rows.values() if v["_dur"] > 1e-3]
    if not vs: continue
    n = len(vs); dur = sum(v["_dur"] for v in vs)/n; cyc = sum(v["GRBM_GUI_ACTIVE"] for v in vs)/n/8; valu = sum(v.get("SQ_INSTS_VALU",0) for v in vs)/n/1024
    print(f"{d:28s} n={n} {dur*1e3:7.3f} ms clock {cyc/dur/1e9:4.2f} GHz VALU/SIMD-cycle {valu/cyc:5.3f} (1 per {cyc/max(valu,1):4.2f})")
PY
grep -h "ms per launch" $R/gpurun_out/s2_clk64_*.log $R/gpurun_out/s2_clkmix_*.log
find $R/gpurun_out/s2_clk* -name "*.csv" -size +3M -delete
