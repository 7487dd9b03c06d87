for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04      ', d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('streamed_1e6_chunks',{}).get('frac'))"
  MRHIP_LIB_PATH=$GRAFT_REPO_ROOT/multirate.jl_amd/abnodyn.so python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 nodyn', d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('streamed_1e6_chunks',{}).get('frac'))"
done
