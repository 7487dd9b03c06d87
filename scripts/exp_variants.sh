R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 4 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "metric|rror|fault|differs" | python -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); print('   single', d['roofline']['achieved'], d['roofline']['avg_launch_ms'], ' streamed', d['streamed_1e6_chunks']['achieved_GBps'], d['streamed_1e6_chunks']['avg_launch_ms'])
    except Exception: print(l[:200])
"; }
{
for i in 1 2; do
run MRHIP_LIB_PATH=$R/build_exp/lib_prime.so
run MRHIP_LIB_PATH=$R/build_exp/lib_base.so
done
} > gpurun_out/exp_prime.log 2>&1
