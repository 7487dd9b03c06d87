R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env MRHIP_DEBUG=1 MRHIP_BENCH_CHUNKED=0 "$@" timeout 300 python scripts/bench_configs.py c2 2>&1 | grep -E "mrhip\]|config" | grep -v "wave [0-9] spends" | sed -e 's/.*"kernel_ms_per_pass": \([0-9.]*\), "wall_ms_per_pass_incl_host": \([0-9.]*\).*/   kernel_ms=\1 wall_ms=\2/' | cut -c1-300 | tail -5; }
{
run MRHIP_PAIR_PROBE=1 MRHIP_PAIR_BPC=2
run MRHIP_PAIR_PROBE=1 MRHIP_PAIR_BPC=1
run MRHIP_PAIR_PROBE=1 MRHIP_PAIR_BPC=1 MRHIP_PAIR_J=4
run MRHIP_PAIR_PROBE=1 MRHIP_PAIR_J=1
run MRHIP_PAIR_BPC=1
run MRHIP_PAIR_BPC=1 MRHIP_PAIR_J=4
run MRHIP_PAIR_BPC=1 MRHIP_PAIR_J=3
} > gpurun_out/exp_c2.log 2>&1
