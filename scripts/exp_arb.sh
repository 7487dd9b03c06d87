R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for c in 4 8; do for t in 256 512 1024; do MRHIP_ARB_TILE=$t MRHIP_ARB_CPL=$c MRHIP_DEBUG=1 timeout 300 python scripts/bench_configs.py c4 2>&1 | grep -E "mrhip\] arb|config" | sed -e 's/.*lds=\([0-9]*\) occ.CU=\([0-9]*\) regs=\([0-9]*\) tile_out=\([0-9]*\).*/   lds=\1 occ=\2 regs=\3 tile=\4/' -e 's/.*"kernel_ms_per_pass": \([0-9.]*\).*"algorithmic_GBps": \([0-9.]*\).*/   ms=\1 GBps=\2/'; done; done
} > gpurun_out/exp_arb.log 2>&1
