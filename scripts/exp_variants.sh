R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python bench.py --steps 3 --warmup 2 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "probe: tile-loop|metric|rror|fault|differs" | sed -e 's/.*"value": \([0-9.]*\).*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   value=\1 GBps=\2 ms=\3/' | cut -c1-300 | tail -3; }
{
run MRHIP_PAIR_LDS_TAPS=1
run MRHIP_PAIR_LDS_TAPS=0
run MRHIP_PAIR_LDS_TAPS=1
run MRHIP_PAIR_LDS_TAPS=0
EXTRA=--no-check run MRHIP_PAIR_LDS_TAPS=1 MRHIP_PAIR_PROBE=1
for v in 1 0; do echo "== C1/C2 LDS_TAPS=$v"; MRHIP_PAIR_LDS_TAPS=$v timeout 300 python scripts/bench_configs.py c1 c2 2>/dev/null | cut -c1-330; done
timeout 600 python -m pytest tests -x -q -m gpu -k "tuned or headline or golden or sweep or config5 or dynamic or fused" 2>&1 | tail -3
} > gpurun_out/exp_ldstaps.log 2>&1
