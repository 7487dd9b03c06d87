#!/bin/bash
# Counter passes for one bench_configs.py row: usage  profile_kernel.sh <tag> <kernel-name-substring> <bench_configs args...>
# (GRAFT_REPO_ROOT must be set; separate --pmc passes, no trace domains mixed in; summaries only are kept)
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"
TAG="${1:?tag}"; KSUB="${2:?kernel substring}"; shift 2
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$R/scripts/bench_configs.py" "$@" > "$OUT/trace.log" 2>&1
P=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" ${EXTRA_PMC:+"$EXTRA_PMC"}; do
  rocprofv3 --output-format csv --pmc $set -d "$OUT/pmc_$P" -o c -- python3 "$R/scripts/bench_configs.py" "$@" > "$OUT/pmc_$P.log" 2>&1
  P=$((P+1))
done
cd "$R" && python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, sys, collections, json
out, ksub = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        res.setdefault("kernel_stats", []).append({k: row[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage") if k in row})
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub not in row["Kernel_Name"]: continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
res["counters_per_dispatch"] = {k: round(v / max(n[k], 1)) for k, v in sorted(tot.items())}
res["dispatches"] = dict(n)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res["counters_per_dispatch"]))
for r in res.get("kernel_stats", [])[:8]: print(r)
PY
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
