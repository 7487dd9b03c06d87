#!/bin/bash
# per-kernel durations of a FIRArbitrary stream that does not repeat (the schedule kernels beside the filter kernel)
# usage (on the GPU box): bash scripts/prof_arb_continuing.sh <tag> [row word ...]
set -e
tag=${1:-arbcont}; shift || true
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof -o run -- python3 $GRAFT_REPO_ROOT/scripts/exp_arb_continuing.py "$@" > $out/run.log 2>&1
f=$(find $out/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $out/kernel_stats.csv
cat $out/run.log | tail -3
column -s, -t < $out/kernel_stats.csv | cut -c1-200 | head -20
