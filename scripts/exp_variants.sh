R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-streamed 2>&1 | grep -E "mrhip\] rational|metric|rror|fault|differs" | sed -e 's/.*"value": \([0-9.]*\).*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   value=\1 GBps=\2 ms=\3/' | sed -e 's/.*grid=\([0-9]*\).*lds=\([0-9]*\).*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   grid=\1 lds=\2 occ=\3 regs=\4 J=\5/' | cut -c1-300 | tail -2; }
{
run MRHIP_PAIR=1
run MRHIP_PAIR_C=8 MRHIP_PAIR_NS=2 MRHIP_PAIR_J=6
run MRHIP_PAIR_C=8 MRHIP_PAIR_NS=2 MRHIP_PAIR_J=5
run MRHIP_PAIR_C=7 MRHIP_PAIR_NS=2 MRHIP_PAIR_J=7
run MRHIP_PAIR_C=6 MRHIP_PAIR_NS=2 MRHIP_PAIR_J=8
run MRHIP_PAIR=1
} > gpurun_out/exp_bigwg.log 2>&1
