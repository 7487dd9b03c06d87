R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
MRHIP_DEBUG=1 python bench.py --no-cpu-baseline 2>&1 | grep -E "mrhip\]|metric" | cut -c1-1900
python scripts/bench_c5_sharded.py 2>/dev/null
python scripts/bench_configs.py c5 c1 2>/dev/null | cut -c1-330
} > gpurun_out/exp_ns2_final.log 2>&1
