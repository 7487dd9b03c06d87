R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for kb in 16 24 32 40 48 60; do MRHIP_DIRECT_TILE_KIB=$kb MRHIP_DEBUG=1 python scripts/bench_configs.py c3b 2>&1 | grep -E "mrhip\] fir|config" | sed -e 's/.*lds=\([0-9]*\) occ.CU=\([0-9]*\) regs=\([0-9]*\) J=\([0-9]*\).*/   lds=\1 occ=\2 regs=\3 J=\4/' -e 's/.*"kernel_ms_per_pass": \([0-9.]*\).*"algorithmic_GBps": \([0-9.]*\).*/   ms=\1 GBps=\2/'; done
} > gpurun_out/exp_dec.log 2>&1
