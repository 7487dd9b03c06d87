R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
echo "=== valu_rate"; timeout 120 scripts/ubench/valu_rate
echo "=== pair_loop_v"; timeout 300 scripts/ubench/pair_loop_v
} > gpurun_out/ubench_pk.log 2>&1
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 600 python bench.py --steps 3 --warmup 1 --samples 50000000 --no-cpu-baseline 2>&1 | grep -E "mrhip\]|metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   occ=\1 regs=\2 J=\3/' | cut -c1-220; }
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do for pk in 0 1; do run MRHIP_PAIR_PK=$pk; done; done
EXTRA="--numerics fused"
runf() { echo "== fused $*"; env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --samples 50000000 --no-cpu-baseline --numerics fused 2>&1 | grep -E "metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | cut -c1-220; }
for pk in 0 1; do runf MRHIP_PAIR_PK=$pk; done
} > gpurun_out/exp_pk.log 2>&1
