#!/bin/bash
# The randomised parity stresses, one after the other, on the GPU box: usage  bash scripts/stress_end_of_round.sh <out-file under gpurun_out/>
# (each prints one summary line; a mismatch makes its script exit non-zero and this one stop)
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"; OUT="$R/gpurun_out/${1:?out file}"
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
cd "$R"
run() { echo "== $*" >> "$OUT"; timeout -k 10 600 "$@" 2>&1 | grep -v "amdgpu.ids" | tail -2 >> "$OUT" || { echo "FAILED: $*" >> "$OUT"; tail -3 "$OUT"; exit 1; }; tail -1 "$OUT"; }
run python tests/stress_random.py --cases 3000 --seed 61
run python tests/stress_random.py --cases 2500 --seed 62
run python tests/stress_random.py --cases 3000 --seed 91 --aligned      # (rows that are multiples of 16 bytes: every kernel's LDS-DMA path)
run python scripts/stress_arb_lane.py 200 63
run python scripts/stress_lane_kernels.py 240 7
run python scripts/stress_schedule.py --cases 1500 --seed 1 --seconds 240
run python scripts/stress_ring.py --cases 400 --seed 6 --seconds 150
run python scripts/stress_blocks.py --cases 3000 --seed 6 --seconds 120
run python scripts/stress_sharded.py --cases 2000 --seed 6 --seconds 120
run python scripts/stress_cascade.py --cases 3000 --seed 6 --seconds 150
run python scripts/stress_graph.py --cases 2000 --seed 6 --seconds 120
run python scripts/stress_multi.py --cases 3000 --seed 6 --seconds 200
run python scripts/stress_host.py --cases 3000 --seed 6 --seconds 120
run python scripts/stress_advance.py --cases 3000 --seed 6 --seconds 200
echo "ALL STRESSES DONE" >> "$OUT"
