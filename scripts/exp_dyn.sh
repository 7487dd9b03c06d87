R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python bench.py --steps 3 --warmup 2 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "mrhip\]|metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*grid=\([0-9]*\).*lds=\([0-9]*\).*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   grid=\1 lds=\2 occ=\3 regs=\4 J=\5/' | cut -c1-220; }
{
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run MRHIP_PAIR_PROBE=1
EXTRA="--no-check"
for j in 2 3 4 6; do run MRHIP_PAIR_J=$j; done
run MRHIP_PAIR_C=4 MRHIP_PAIR_J=4
run MRHIP_PAIR_C=2 MRHIP_PAIR_J=6
EXTRA="--numerics fused --no-check"; run A=1
} > gpurun_out/exp_dyn.log 2>&1
