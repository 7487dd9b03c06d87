#!/bin/bash
# Copies what scripts/profile_round.sh left under gpurun_out/<tag>/ into profiles/<dest>/ (the judged copy): the rocprofv3 kernel-stats CSV
# of every run, the bench / bench_configs output taken under the profiler, and the summary.
#   bash scripts/collect_profiles.sh r06/round1 r06
set -eu
SRC="gpurun_out/${1:?tag under gpurun_out}"; DST="profiles/${2:?directory under profiles}"
mkdir -p "$DST"
for d in "$SRC"/prof_*/; do
  n=$(basename "$d"); n=${n#prof_}
  f=$(find "$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$DST/rocprof_${n}_kernel_stats.csv"
done
for f in "$SRC"/*_under_rocprof.json "$SRC"/*_under_rocprof.jsonl; do [ -f "$f" ] && cp "$f" "$DST/"; done
[ -f "$SRC/summary.json" ] && cp "$SRC/summary.json" "$DST/rocprof_summary.json"
ls "$DST"
