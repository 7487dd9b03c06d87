#!/bin/bash
# A/B on ONE box: the product library against multirate.jl_amd/mrhip_exp.so (make EXP=1 EXPFLAGS=...), alternating.  usage: ab_exp.sh <rows...>
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"
for i in 1 2 3; do
  for lib in product exp; do
    if [ $lib = exp ]; then export MRHIP_LIB_PATH="$R/multirate.jl_amd/mrhip_exp.so"; else unset MRHIP_LIB_PATH; fi
    python scripts/bench_configs.py "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['config'][:44].ljust(44), d['kernel'][:22].ljust(22), d['kernel_ms_per_pass'], d['frac_of_8TBps'])"
  done
done
