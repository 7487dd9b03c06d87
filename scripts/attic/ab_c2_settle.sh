for s in 0 60 0 40 30 60 100; do
  BENCH_SETTLE_MS=$s timeout -k 10 300 python scripts/bench_configs.py c2 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('settle $s', d['config'][:40], d['kernel_ms_per_pass'], d['wall_ms_per_pass_incl_host'], d['frac_of_8TBps'], d['untimed_passes'], d['untimed_ms'])
"
done
