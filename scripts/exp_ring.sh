R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 600 python bench.py --steps 3 --warmup 1 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "mrhip\]|metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*grid=\([0-9]*\).*lds=\([0-9]*\).*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   grid=\1 lds=\2 occ=\3 regs=\4 J=\5/' | cut -c1-220; }
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
run A=1; run A=1
EXTRA="--no-check"
run MRHIP_PS_ABLATE=3
run MRHIP_PS_ABLATE=7
for c in 2 3 4; do for r in 2 3 4; do run MRHIP_PAIR_C=$c MRHIP_PAIR_ROUNDS=$r; done; done
run MRHIP_PAIR_C=4 MRHIP_PAIR_ROUNDS=2 MRHIP_PS_ABLATE=3
run MRHIP_PAIR_C=4 MRHIP_PAIR_ROUNDS=2 MRHIP_PS_ABLATE=7
EXTRA="--numerics fused --no-check"; run A=1
} > gpurun_out/exp_ring.log 2>&1
