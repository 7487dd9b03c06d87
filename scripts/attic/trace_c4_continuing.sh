#!/bin/bash
# Kernel trace of BASELINE config 4 on a continuing stream: where does the schedule of call i + 1 run relative to the filter kernel of call i?
# usage (GPU box): bash scripts/attic/trace_c4_continuing.sh <tag> [ENV=VALUE ...]   (TRACE_BENCH=1: the row inside bench.py's full table)
R="${GRAFT_REPO_ROOT:?}"; TAG="$1"; shift
OUT="$R/gpurun_out/r06/trace_$TAG"; mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
if [ "${TRACE_BENCH:-0}" = 1 ]; then
  rocprofv3 --output-format csv --kernel-trace -d "$OUT" -o t -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-streamed > "$OUT/out.jsonl" 2> "$OUT/err.log"
else
  rocprofv3 --output-format csv --kernel-trace -d "$OUT" -o t -- python3 "$R/scripts/bench_configs.py" c4 > "$OUT/out.jsonl" 2> "$OUT/err.log"
fi
cd "$R" && python3 - "$OUT" <<'P'
import csv, glob, sys, json
out = sys.argv[1]
fn = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
lane = [i for i, r in enumerate(rows) if "arb_lane_kernel<false" in r["Kernel_Name"]]
t0 = int(rows[lane[-6]]["Start_Timestamp"])
for r in rows[lane[-6] - 2:lane[-1] + 8]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6; e = (int(r["End_Timestamp"]) - t0) / 1e6
    nm = r["Kernel_Name"]; nm = nm[nm.find("namespace)::") + 12:] if "namespace)::" in nm else nm; nm = nm.split("(")[0][:28]
    if e - s > 0.02 or "lane" in nm: print(f"{s:9.3f} {e:9.3f} {e - s:7.3f} q={r['Queue_Id']} {nm}")
for ln in open(out + "/out.jsonl"):
    if ln.startswith("{") and '"C4 ' in ln:
        d = json.loads(ln); print("row:", d["kernel_ms_per_pass"], d["wall_ms_per_pass_incl_host"], d.get("kernel_ms_with_schedule_memo"))
P
rm -f "$fn"; find "$OUT" -name "*.csv" -size +1M -delete
