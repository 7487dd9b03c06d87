R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python bench.py --steps 3 --warmup 2 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "mrhip\] rat|metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*grid=\([0-9]*\).*lds=\([0-9]*\).*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   grid=\1 lds=\2 occ=\3 regs=\4 J=\5/' | cut -c1-220; }
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
run A=1
for c in 2 3 4; do for ns in 4 6 8; do for j in 1 2 3; do run MRHIP_PAIR_C=$c MRHIP_PAIR_NS=$ns MRHIP_PAIR_J=$j; done; done; done
} > gpurun_out/exp_ns.log 2>&1
