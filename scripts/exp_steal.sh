R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python bench.py --steps 3 --warmup 2 --samples 50000000 --no-cpu-baseline $EXTRA 2>&1 | grep -E "mrhip\] probe: tile|mrhip\] rational|metric|rror|differs|fault" | sed -e 's/.*"value": \([0-9.]*\).*"achieved": \([0-9.]*\).*"avg_launch_ms": \([0-9.]*\).*/   value=\1 GBps=\2 ms=\3/' | sed -e 's/.*grid=\([0-9]*\).*lds=\([0-9]*\).*occ.CU=\([0-9]*\) regs=\([0-9]*\).*J=\([0-9]*\).*/   grid=\1 lds=\2 occ=\3 regs=\4 J=\5/' | cut -c1-300 | tail -3; }
{
run MRHIP_PAIR=1
run MRHIP_PAIR_J=4 MRHIP_PAIR_BPC=4
run MRHIP_PAIR=1
run MRHIP_PAIR_J=4 MRHIP_PAIR_BPC=4
} > gpurun_out/exp_occ6.log 2>&1
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t.log
python scripts/bench_configs.py c1 c2 c5 > gpurun_out/configs11.jsonl 2>gpurun_out/configs11.err
