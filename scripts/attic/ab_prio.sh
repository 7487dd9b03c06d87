# A/B: loader wave at s_setprio 3 (libmrhip_prio.so, -DMRHIP_LOADER_PRIO=3) against the product library, alternating on one box
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'one call', d['roofline']['frac'], 'avg ms', d['roofline']['avg_launch_ms'], 'streamed', d.get('streamed_1e6_chunks',{}).get('frac'))"; }
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-configs 2>/dev/null | line "product "
  MRHIP_LIB_PATH=$GRAFT_REPO_ROOT/multirate.jl_amd/libmrhip_prio.so python bench.py --no-cpu-baseline --no-configs 2>/dev/null | line "prio 3  "
done
