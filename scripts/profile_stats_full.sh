# rocprofv3 kernel stats of the CHUNKED stream (bench.py --chunk 1000000; profiles/r01c), for the duration-agreement check.
# The default (single-call) command is profiled by scripts/profile_headline.sh.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
rm -rf $OUT/prof_full
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_full -o full -- python3 $R/bench.py --chunk 1000000 > $OUT/prof_full_bench.json 2> $OUT/prof_full.log
cd $R && python3 - <<'PY'
import csv, glob, json, statistics
rows = list(csv.DictReader(open(glob.glob('gpurun_out/prof_full/**/full_kernel_trace.csv', recursive=True)[0])))
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if 'rational_pair_kernel' in r['Kernel_Name']]
n = len(d)
timed = d[100:]          # the first 100 launches are the warm-up step
out = {"kernel": "rational_pair_kernel<24,false,1>", "launches": n, "avg_us_all": sum(d) / n / 1e3,
       "avg_us_timed_region": sum(timed) / len(timed) / 1e3, "median_us_timed_region": statistics.median(timed) / 1e3,
       "min_us": min(d) / 1e3, "max_us": max(d) / 1e3}
json.dump(out, open('gpurun_out/prof_full_summary.json', 'w'), indent=1)
print(out)
PY
find gpurun_out/prof_full -name "*.csv" -size +3M -delete
