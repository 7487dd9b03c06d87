R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests -x -q -m gpu -k "not tuned_and_generic" 2>&1 | tail -3
MRHIP_DEBUG=1 python bench.py --steps 3 --warmup 1 --samples 50000000 --no-cpu-baseline 2>&1 | grep -E "mrhip\] rat|metric" | cut -c1-400
MRHIP_DEBUG=1 python scripts/bench_configs.py c5 2>&1 | grep -E "mrhip\] rat|config" | cut -c1-400
for c in 2 3 4; do for j in 1 2 3 4; do MRHIP_PAIR_C=$c MRHIP_PAIR_J=$j MRHIP_DEBUG=1 python scripts/bench_configs.py c5 2>&1 | grep -E "mrhip\] rat|config" | sed -e 's/.*lds=\([0-9]*\) occ.CU=\([0-9]*\) regs=\([0-9]*\) c=\([0-9]*\).*J=\([0-9]*\).*/   lds=\1 occ=\2 regs=\3 c=\4 J=\5/' -e 's/.*"kernel_ms_per_pass": \([0-9.]*\).*"algorithmic_GBps": \([0-9.]*\).*/   ms=\1 GBps=\2/'; done; done
} > gpurun_out/exp_c5.log 2>&1
