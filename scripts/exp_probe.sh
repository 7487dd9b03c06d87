R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 600 python bench.py --steps 2 --warmup 2 --samples 20000000 --no-cpu-baseline --no-check $EXTRA 2>&1 | grep -E "mrhip\]|metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*grid=\([0-9]*\).*lds=\([0-9]*\).*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   grid=\1 lds=\2 occ=\3 regs=\4 J=\5/' | cut -c1-400 | tail -12; }
{
run MRHIP_PAIR_PROBE=1 MRHIP_PAIR_C=4 MRHIP_PAIR_BPC=3
run MRHIP_PAIR_PROBE=1 MRHIP_PAIR_C=3
} > gpurun_out/exp_probe4.log 2>&1
