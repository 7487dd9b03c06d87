R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 120 python bench.py --steps 3 --warmup 2 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "mrhip\]|metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*grid=\([0-9]*\).*lds=\([0-9]*\).*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   grid=\1 lds=\2 occ=\3 regs=\4 J=\5/' | cut -c1-400 | grep -E "GBps|wave [0-9] spends" | tail -7; }
{
for rep in 1 2; do
run MRHIP_PAIR_PRIO=0
run MRHIP_PAIR_PRIO=1
run MRHIP_PAIR_PRIO=2
done
run MRHIP_PAIR_PRIO=1 MRHIP_PAIR_PROBE=1
run MRHIP_PAIR_PRIO=1 MRHIP_PAIR_BPC=3
run MRHIP_PAIR_PRIO=1 MRHIP_PAIR_C=3
} > gpurun_out/exp_prio.log 2>&1
