R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python bench.py --steps 3 --warmup 2 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "probe: in-kernel|probe: tile-loop|metric|rror|fault|differs" | sed -e 's/.*"value": \([0-9.]*\).*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   value=\1 GBps=\2 ms=\3/' | cut -c1-300 | tail -5; }
{
run MRHIP_PAIR_NT=0
run MRHIP_PAIR_NT=1
run MRHIP_PAIR_NT=0
run MRHIP_PAIR_NT=1
EXTRA=--no-check run MRHIP_PAIR_NT=1 MRHIP_PAIR_PROBE=1
} > gpurun_out/exp_nt.log 2>&1
