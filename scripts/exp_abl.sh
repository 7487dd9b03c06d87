R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 600 python bench.py --steps 3 --warmup 1 --samples 50000000 --no-cpu-baseline --no-check $EXTRA 2>&1 | grep -E "mrhip\]|metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | sed -e 's/.*grid=\([0-9]*\).*lds=\([0-9]*\).*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   grid=\1 lds=\2 occ=\3 regs=\4 J=\5/' | cut -c1-220; }
{
for c in 2 4; do
export MRHIP_PAIR_C=$c MRHIP_PAIR_ROUNDS=2
run MRHIP_PS_ABLATE=7
run MRHIP_PS_ABLATE=15
run MRHIP_PS_ABLATE=23
run MRHIP_PS_ABLATE=39
run MRHIP_PS_ABLATE=47
run MRHIP_PS_ABLATE=55
run MRHIP_PS_ABLATE=63
done
} > gpurun_out/exp_abl.log 2>&1
