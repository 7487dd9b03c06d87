R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 200 python bench.py --steps 4 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "metric|rror|fault|differs" | python -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); print('   single', d['roofline']['achieved'], d['roofline']['avg_launch_ms'], ' streamed', d['streamed_1e6_chunks']['achieved_GBps'], d['streamed_1e6_chunks']['avg_launch_ms'])
    except Exception: print(l[:200])
"; }
{
echo "== smoke-sized correctness first (bounded)"
MRHIP_LIB_PATH=$R/build_exp/lib_desc.so timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
MRHIP_LIB_PATH=$R/build_exp/lib_desc.so timeout 300 python -m pytest tests -x -q -m gpu -k "headline or dynamic or long_launch or config5 or fused or golden or chunked or graph" 2>&1 | tail -2
for i in 1 2; do
run MRHIP_LIB_PATH=$R/build_exp/lib_desc.so
run MRHIP_LIB_PATH=$R/build_exp/lib_base.so
done
} > gpurun_out/exp_desc.log 2>&1
