R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check $EXTRA 2>&1 | grep -E "metric|rror|differs" | sed -e 's/.*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   GBps=\1 ms=\2/' | cut -c1-220; }
{
for ab in 0 3; do
for ck in 250000 500000 1000000 2000000 4000000 8000000; do EXTRA="--samples 32000000 --chunk $ck"; run MRHIP_PS_ABLATE=$ab; done
done
} > gpurun_out/exp_chunk.log 2>&1
