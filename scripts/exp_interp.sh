R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{


for w in 2 4 6; do for j in 4 8 12 16; do MRHIP_INTERP_WAVES=$w MRHIP_INTERP_J=$j MRHIP_DEBUG=1 python scripts/bench_configs.py c3a 2>&1 | grep -E "mrhip\] int|config" | sed -e 's/.*lds=\([0-9]*\) occ.CU=\([0-9]*\) regs=\([0-9]*\) CP=\([0-9]*\).*J=\([0-9]*\).*/   lds=\1 occ=\2 regs=\3 CP=\4 J=\5/' -e 's/.*"kernel_ms_per_pass": \([0-9.]*\).*"algorithmic_GBps": \([0-9.]*\).*/   ms=\1 GBps=\2/'; done; done

} > gpurun_out/exp_interp.log 2>&1
