R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 "$@" timeout 300 python scripts/bench_configs.py c3a 2>&1 | grep -E "mrhip\]|config" | sed -e 's/.*"kernel_ms_per_pass": \([0-9.]*\).*"algorithmic_GBps": \([0-9.]*\).*/   kernel_ms=\1 GBps=\2/' | cut -c1-250; }
{
run MRHIP_INTERP=1
run MRHIP_INTERP_NS=2
run MRHIP_INTERP_NS=2 MRHIP_INTERP_J=12
run MRHIP_INTERP_NS=2 MRHIP_INTERP_J=16
run MRHIP_INTERP_J=4
run MRHIP_INTERP=1
} > gpurun_out/exp_interp_ns2.log 2>&1
