#!/bin/bash
# PMC passes (separate runs, no trace domains mixed in) of the three kernels VERDICT round 3 asks about -- C3a (rational_opair,
# M = 1), C3b (fir_stream) and the README's mixed precision (rational_opair, Float64 arithmetic on Float32 samples) -- in STRICT
# and in FUSED numerics.  Summaries land in gpurun_out/r04_item7/<row>_<numerics>/summary.json.
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"
for row in c3a c3b xmix64; do
  case $row in c3b) K=fir_stream_kernel;; *) K=rational_opair_kernel;; esac
  bash "$R/scripts/profile_kernel.sh" "r04_item7/${row}_strict" $K $row > "$R/gpurun_out/r04_item7_${row}_strict.log" 2>&1
  echo "done $row strict"
  bash "$R/scripts/profile_kernel.sh" "r04_item7/${row}_fused" $K $row --numerics fused > "$R/gpurun_out/r04_item7_${row}_fused.log" 2>&1
  echo "done $row fused"
done
