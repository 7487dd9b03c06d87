# A/B in ONE library: the headline's kernel with its arguments left alone (PLAIN instantiation, product) vs the instantiation that may take a
# call record / stream descriptors (MRHIP_OPAIR_PLAIN=0), alternating on one box
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'one call', d['roofline']['frac'], 'avg ms', d['roofline']['avg_launch_ms'], 'streamed', d.get('streamed_1e6_chunks',{}).get('frac'))"; }
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-configs 2>/dev/null | line "plain    "
  MRHIP_OPAIR_PLAIN=0 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | line "dyn-ready"
done
