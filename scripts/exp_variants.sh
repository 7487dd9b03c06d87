R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python bench.py --steps 3 --warmup 2 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "metric|rror|fault|differs" | sed -e 's/.*"value": \([0-9.]*\).*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   value=\1 GBps=\2 ms=\3/' | cut -c1-300 | tail -3; }
{
for i in 1 2; do
run MRHIP_LIB_PATH=$R/build_exp/lib_paired.so
run MRHIP_LIB_PATH=$R/build_exp/lib_single.so
done
run MRHIP_LIB_PATH=$R/build_exp/lib_paired.so MRHIP_PS_ABLATE=3
run MRHIP_LIB_PATH=$R/build_exp/lib_single.so MRHIP_PS_ABLATE=3
EXTRA="--numerics fused" run MRHIP_LIB_PATH=$R/build_exp/lib_paired.so
EXTRA="--numerics fused" run MRHIP_LIB_PATH=$R/build_exp/lib_single.so
} > gpurun_out/exp_waits.log 2>&1
