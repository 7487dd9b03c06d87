R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python scripts/bench_configs.py x160 xf64 2>&1 | grep -E "mrhip\]|config" | sed -e 's/.*"kernel_ms_per_pass": \([0-9.]*\).*"algorithmic_GBps": \([0-9.]*\).*/   kernel_ms=\1 GBps=\2/' | cut -c1-200; }
{
run MRHIP_PS_J=0
run MRHIP_PS_J=4
run MRHIP_PS_J=12
run MRHIP_PS_J=16
run MRHIP_PS_BPC=2
run MRHIP_PS_BPC=4
} > gpurun_out/exp_ps.log 2>&1
