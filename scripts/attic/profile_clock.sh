#!/bin/bash
# The clock the chip holds under each kernel, and how busy its vector ALUs are at that clock: one counter pass
# (GRBM_GUI_ACTIVE, SQ_INSTS_VALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES) over scripts/bench_configs.py rows and the bench's launch.
# effective clock = GRBM_GUI_ACTIVE / 8 / dispatch duration (MI355X_MICROARCH.md, DVFS give-back);
# VALU instructions per SIMD-cycle = SQ_INSTS_VALU / 1024 / (GRBM_GUI_ACTIVE / 8).
# usage: bash scripts/profile_clock.sh <tag> [config names...]      (GRAFT_REPO_ROOT must be set)
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
TAG="${1:?usage: profile_clock.sh <tag> [configs...]}"
shift
CFG="${*:-c3a c3b xstd xf64 xmix x160 c5 c4}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/clk_cfg" "$OUT/clk_bench"
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/clk_cfg" -o clk -- python3 "$R/scripts/bench_configs.py" $CFG > "$OUT/clk_cfg.log" 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/clk_bench" -o clk -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-streamed > "$OUT/clk_bench.log" 2>&1
cd "$R" && python3 - "$OUT" > "$OUT/clock_summary.txt" <<'PY'
import csv, glob, os, re, sys, collections
out = sys.argv[1]
csv.field_size_limit(1 << 30)
for d in ("clk_bench", "clk_cfg"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not re.search(r"mrhip", k): continue
            key = (r["Dispatch_Id"], re.sub(r"^.*mrhip::(\(anonymous namespace\)::)?", "", k)[:70], r["Grid_Size"])
            rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["_dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    agg = collections.defaultdict(list)
    for (did, name, grid), v in rows.items():
        if v["_dur"] < 50e-6 or "GRBM_GUI_ACTIVE" not in v: continue          # the quotient reads high on short dispatches
        agg[(name, grid)].append(v)
    print(f"== {d}")
    for (name, grid), vs in sorted(agg.items()):
        n = len(vs)
        dur = sum(v["_dur"] for v in vs) / n
        cyc = sum(v["GRBM_GUI_ACTIVE"] for v in vs) / n / 8
        valu = sum(v.get("SQ_INSTS_VALU", 0) for v in vs) / n / 1024
        print(f"{name:72s} grid {grid:>8s} n={n:3d}  {dur * 1e3:9.4f} ms  clock {cyc / dur / 1e9:5.2f} GHz  VALU instr per SIMD-cycle {valu / cyc:5.3f}  (one per {cyc / max(valu, 1):4.2f} cycles)")
PY
find "$OUT" -name "*.csv" -size +3M -delete; find "$OUT" -name "*.db" -delete
cat "$OUT/clock_summary.txt"
