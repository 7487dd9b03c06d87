#!/bin/bash
# round-2 experiment: opair kernel correctness + where it stands (x160, 3//2, and on the headline shape with the position-pair kernel off)
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd "$R" || exit 1
OUT="$R/gpurun_out/r2b"; mkdir -p "$OUT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > "$OUT/pytest.txt"; tail -8 "$OUT/pytest.txt"
python scripts/bench_configs.py x160 x32 xf64 xmix c2 c2s > "$OUT/configs.jsonl" 2> "$OUT/err.txt"; cat "$OUT/configs.jsonl"
MRHIP_OPAIR=0 python scripts/bench_configs.py x160 x32 > "$OUT/configs_noopair.jsonl" 2>> "$OUT/err.txt"; cat "$OUT/configs_noopair.jsonl"
for c in 4 6; do
  MRHIP_DEBUG=1 MRHIP_PAIR=0 MRHIP_OPAIR_C=$c python bench.py --no-cpu-baseline --steps 3 > "$OUT/headline_opair_c$c.json" 2>> "$OUT/err.txt"; python - "$OUT/headline_opair_c$c.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d["kernel"], d["roofline"]["frac"], d.get("streamed_1e6_chunks",{}).get("frac"))
PY
done
grep "mrhip" "$OUT/err.txt" | head
