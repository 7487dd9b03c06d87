#!/bin/bash
# More seeds of the randomised parity stress than any round has run (experiments.md K: a new seed found a three-round-old bug):
# usage  bash scripts/stress_more_seeds.sh <out-file under gpurun_out/> [a|b]   (a: the ten plain seeds, b: --big / --async-mix / --arb; each part ~18 min)
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"; OUT="$R/gpurun_out/${1:?out file}"
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
cd "$R"
run() { echo "== $*" >> "$OUT"; timeout -k 10 400 "$@" 2>&1 | grep -v "amdgpu.ids" | grep "MISMATCH\|^cases\|mismatches" | tail -4 >> "$OUT"; tail -1 "$OUT" | cut -c1-120; }
PART="${2:-all}"
if [ "$PART" != b ]; then
for s in 71 72 73 74 75 76 77 78 79 80; do run python tests/stress_random.py --cases 3000 --seed $s --seconds 120; done
fi
if [ "$PART" != a ]; then
for s in 81 82 83; do run python tests/stress_random.py --cases 400 --seed $s --big --seconds 150; done
for s in 84 85; do run python tests/stress_random.py --cases 2000 --seed $s --async-mix 0.5 --seconds 120; done
for s in 86 87; do run python tests/stress_random.py --cases 2000 --seed $s --arb 0.6 --seconds 120; done
fi
echo "DONE" >> "$OUT"
