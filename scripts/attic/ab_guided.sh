# A/B: guided grab sizes over the last round of a scheduling group (pair_loader.h), per process through the environment
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'one call', d['roofline']['frac'], 'avg ms', d['roofline']['avg_launch_ms'], 'streamed', d.get('streamed_1e6_chunks',{}).get('frac'))"; }
for rep in 1 2; do
  MRHIP_GRAB_GUIDED=0 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | line "off      "
  for d in 2 4 8 64; do
    MRHIP_GRAB_MIN_DIV=$d python bench.py --no-cpu-baseline --no-configs 2>/dev/null | line "min J/$d "
  done
done
for g in 0 1; do
  echo "guided=$g"
  MRHIP_GRAB_GUIDED=$g python scripts/bench_configs.py c2 xmix64 ms c3b c5 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('  ', d['config'][:70], d.get('kernel_ms_per_pass'), d.get('frac_of_8TBps'))"
done
